// ppp_api.hip -- the extern "C" entry points declared in include/ppp_mi355x.h.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "ppp_kernels.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    return fail(PPP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// largest float <= t  /  smallest float >= t
float round_down_f32(double t) {
    float f = (float)t;
    if ((double)f > t) f = std::nextafterf(f, -INFINITY);
    return f;
}
float round_up_f32(double t) {
    float f = (float)t;
    if ((double)f < t) f = std::nextafterf(f, INFINITY);
    return f;
}

int make_geo(const ppp_params *p, ppp::Geo *G) {
    if (!p) return fail(PPP_ERR_INVALID_ARG, "params is NULL");
    if (p->abi_version != PPP_ABI_VERSION)
        return fail(PPP_ERR_INVALID_ARG, "abi_version %d != %d", p->abi_version, PPP_ABI_VERSION);
    if (p->Z <= 0 || p->Y <= 0 || p->X <= 0)
        return fail(PPP_ERR_INVALID_ARG, "bad volume %d x %d x %d", p->Z, p->Y, p->X);
    if (p->pz <= 0 || p->py <= 0 || p->px <= 0 || !(p->pz & 1) || !(p->py & 1) || !(p->px & 1))
        return fail(PPP_ERR_INVALID_ARG, "patch shape must be positive and odd (%d,%d,%d)", p->pz,
                    p->py, p->px);
    if (!(p->th > 0.0 && p->th < 1.0)) return fail(PPP_ERR_INVALID_ARG, "th must be in (0,1)");
    if ((long long)p->Z * p->Y * p->X >= (1ll << 31))
        return fail(PPP_ERR_UNSUPPORTED, "volumes of 2^31 voxels or more are not supported");
    ppp::Geo g;
    memset(&g, 0, sizeof(g));
    g.Z = p->Z; g.Y = p->Y; g.X = p->X;
    g.pz = p->pz; g.py = p->py; g.px = p->px;
    g.rz = p->pz / 2; g.ry = p->py / 2; g.rx = p->px / 2;
    g.C = p->pz * p->py * p->px;
    g.mid = g.C / 2;
    g.V = (long long)p->Z * p->Y * p->X;
    g.th_gt = round_down_f32(p->th);
    double bg;
    switch (p->bg_rule) {
    case PPP_BG_INV_TH: bg = p->thi; break;
    case PPP_BG_HALF_TH: bg = p->th / 2; break;
    case PPP_BG_LESS_THAN_TH: bg = p->th; break;
    default: return fail(PPP_ERR_INVALID_ARG, "how is bg defined for vote instances? (bg_rule %d)", p->bg_rule);
    }
    g.bg_lt = round_up_f32(bg);
    g.th_rn = (float)p->th;
    g.th2 = p->th * p->th;
    g.den = 1.0 - p->th * p->th;
    if (p->value_rule < PPP_VAL_COUNT || p->value_rule > PPP_VAL_NORM_PROB_PRODUCT)
        return fail(PPP_ERR_INVALID_ARG, "bad value_rule %d", p->value_rule);
    g.value_rule = p->value_rule;
    g.use_overlap = p->use_overlap != 0;
    g.normalise = p->normalise != 0;
    g.norm_rank = p->norm_rank != 0;
    g.count_pos_neg = p->count_pos_neg != 0;
    g.norm_aff = p->norm_aff != 0;
    g.layout = p->cons_layout;
    const ppp_box &b = p->cons_box;
    if (b.z0 < 0 || b.y0 < 0 || b.x0 < 0 || b.z1 > p->Z || b.y1 > p->Y || b.x1 > p->X ||
        b.z1 <= b.z0 || b.y1 <= b.y0 || b.x1 <= b.x0)
        return fail(PPP_ERR_INVALID_ARG, "cons_box outside the volume or empty");
    if (g.layout == PPP_CONS_REFERENCE) {
        if (b.z0 || b.y0 || b.x0 || b.z1 != p->Z || b.y1 != p->Y || b.x1 != p->X)
            return fail(PPP_ERR_INVALID_ARG, "reference layout needs cons_box = whole volume");
    } else if (g.layout != PPP_CONS_COMPACT && g.layout != PPP_CONS_VOXEL_MAJOR) {
        return fail(PPP_ERR_INVALID_ARG, "bad cons_layout %d", p->cons_layout);
    }
    g.bz0 = b.z0; g.by0 = b.y0; g.bx0 = b.x0;
    g.bZ = b.z1 - b.z0; g.bY = b.y1 - b.y0; g.bX = b.x1 - b.x0;
    g.BV = (long long)g.bZ * g.bY * g.bX;
    g.cz0 = g.bz0; g.cy0 = g.by0; g.cx0 = g.bx0;
    g.cZ = g.bZ; g.cY = g.bY; g.cX = g.bX;
    g.nsy = 2 * p->py; g.nsx = 2 * p->px;
    g.wy = 2 * p->py - 1; g.wx = 2 * p->px - 1;
    g.n_planes = ((2 * p->pz - 1) * g.wy * g.wx - 1) / 2;
    g.oz = p->origin_z; g.oy = p->origin_y; g.ox = p->origin_x;
    if (p->ring_z < 0 || (p->ring_z > 0 && (g.layout != PPP_CONS_VOXEL_MAJOR || g.bZ > p->ring_z || p->origin_z < 0)))
        return fail(PPP_ERR_INVALID_ARG, "ring_z: VOXEL_MAJOR rows only, cons_box at most ring_z slices thick");
    g.ring = p->ring_z;
    g.pred_clean = p->pred_clean == 1 ? 1 : 0;
    if (p->rank_tile < 0 || p->rank_tile > 3) return fail(PPP_ERR_INVALID_ARG, "rank_tile must be 0 .. 3");
    g.rank_tile = p->rank_tile;
    *G = g;
    return PPP_OK;
}

int need_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(PPP_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    return PPP_OK;
}

int check_dtype(int dt) {
    if (dt != PPP_F32 && dt != PPP_F16) return fail(PPP_ERR_INVALID_ARG, "bad pred dtype %d", dt);
    return PPP_OK;
}

}  // namespace

#define PPP_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != PPP_OK) return rc_; \
    } while (0)

namespace ppp {
static int g_env_epoch = 0;
int env_epoch() { return g_env_epoch; }
void env_reload() { ++g_env_epoch; }
}  // namespace ppp

extern "C" {

int ppp_abi_version(void) { return PPP_ABI_VERSION; }
const char *ppp_last_error(void) { return g_err; }
const char *ppp_consensus_kernel_name(void) { return ppp::last_consensus_kernel(); }

void ppp_reload_env(void) { ppp::env_reload(); }

int ppp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int64_t ppp_cons_planes(const ppp_params *p) {
    if (!p) return -1;
    if (p->cons_layout == PPP_CONS_REFERENCE)
        return (int64_t)(p->pz > 1 ? 2 * p->pz : p->pz) * (2 * p->py) * (2 * p->px);
    const int64_t w = (int64_t)(2 * p->pz - 1) * (2 * p->py - 1) * (2 * p->px - 1);
    return p->cons_layout == PPP_CONS_VOXEL_MAJOR ? w : (w - 1) / 2;
}

int64_t ppp_cons_elems(const ppp_params *p) {
    if (!p) return -1;
    const ppp_box &b = p->cons_box;
    return ppp_cons_planes(p) * (int64_t)(b.z1 - b.z0) * (b.y1 - b.y0) * (b.x1 - b.x0);
}

int ppp_consensus(const void *d_pred, int pred_dtype, const uint8_t *d_overlap, float *d_cons,
                  float *d_count, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_pred || (!d_cons && !d_count)) return fail(PPP_ERR_INVALID_ARG, "NULL pred / outputs");
    if (G.use_overlap && !d_overlap) return fail(PPP_ERR_INVALID_ARG, "use_overlap set but d_overlap is NULL");
    if (G.layout == PPP_CONS_VOXEL_MAJOR && (!ppp::consensus_v3_supported(G) || d_count || !d_cons))
        return fail(PPP_ERR_UNSUPPORTED, "ppp_consensus writes VOXEL_MAJOR only with the packed kernel (TH = 0.5, normalised "
                                         "product, px in {3,5,7,9}; no counts): see ppp_consensus_writes_voxel_major");
    if (G.ring) return fail(PPP_ERR_UNSUPPORTED, "ring_z: rows of a ring are written by ppp_consensus_part");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_consensus(d_pred, pred_dtype, d_overlap, d_cons, d_count, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_consensus");
}

int ppp_consensus_rows(const void *d_pred, int pred_dtype, const uint8_t *d_overlap, float *d_cons,
                       const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_pred || !d_cons) return fail(PPP_ERR_INVALID_ARG, "NULL pred / outputs");
    if (G.use_overlap && !d_overlap) return fail(PPP_ERR_INVALID_ARG, "use_overlap set but d_overlap is NULL");
    if (G.layout != PPP_CONS_VOXEL_MAJOR || !ppp::consensus_v3_supported(G))
        return fail(PPP_ERR_UNSUPPORTED, "ppp_consensus_rows writes VOXEL_MAJOR rows with the packed kernel only "
                                         "(see ppp_consensus_writes_voxel_major)");
    if (G.ring) return fail(PPP_ERR_UNSUPPORTED, "ring_z: rows of a ring are written by ppp_consensus_part");
    PPP_TRY(need_device());
    G.vm_open = 1;
    hipError_t e = ppp::launch_consensus(d_pred, pred_dtype, d_overlap, d_cons, nullptr, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_consensus_rows");
}

int ppp_consensus_part(const void *d_pred, int pred_dtype, const uint8_t *d_overlap, float *d_cons,
                       const ppp_params *p, const ppp_box *part, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_pred || !d_cons || !part) return fail(PPP_ERR_INVALID_ARG, "NULL pred / output / part");
    if (G.use_overlap && !d_overlap) return fail(PPP_ERR_INVALID_ARG, "use_overlap set but d_overlap is NULL");
    if ((G.layout != PPP_CONS_VOXEL_MAJOR && G.layout != PPP_CONS_COMPACT) || !ppp::consensus_v3_supported(G))
        return fail(PPP_ERR_UNSUPPORTED, "ppp_consensus_part writes COMPACT planes or VOXEL_MAJOR rows with the packed "
                                         "kernel only (see ppp_consensus_writes_voxel_major)");
    const ppp_box &b = p->cons_box;
    if (part->z0 < b.z0 || part->y0 < b.y0 || part->x0 < b.x0 || part->z1 > b.z1 || part->y1 > b.y1 ||
        part->x1 > b.x1 || part->z1 <= part->z0 || part->y1 <= part->y0 || part->x1 <= part->x0)
        return fail(PPP_ERR_INVALID_ARG, "part must be a non-empty sub-box of cons_box");
    PPP_TRY(need_device());
    G.vm_open = 1;
    G.cz0 = part->z0; G.cy0 = part->y0; G.cx0 = part->x0;
    G.cZ = part->z1 - part->z0; G.cY = part->y1 - part->y0; G.cX = part->x1 - part->x0;
    hipError_t e = ppp::launch_consensus_part(d_pred, pred_dtype, d_overlap, d_cons, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_consensus_part");
}

int ppp_consensus_writes_voxel_major(const ppp_params *p) {
    ppp::Geo G;
    if (!p || make_geo(p, &G) != PPP_OK) return 0;
    return ppp::consensus_v3_supported(G) ? 1 : 0;
}

int ppp_rank_patches(const void *d_pred, int pred_dtype, const float *d_cons,
                     const uint8_t *d_overlap, float *d_score, const ppp_box *score_box,
                     const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_pred || !d_cons || !d_score) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.use_overlap && !d_overlap) return fail(PPP_ERR_INVALID_ARG, "use_overlap set but d_overlap is NULL");
    if (G.layout == PPP_CONS_VOXEL_MAJOR) return fail(PPP_ERR_UNSUPPORTED, "ppp_rank_patches reads COMPACT or REFERENCE layout");
    ppp_box sb = {0, 0, 0, p->Z, p->Y, p->X};
    if (score_box) sb = *score_box;
    if (sb.z0 < 0 || sb.y0 < 0 || sb.x0 < 0 || sb.z1 > p->Z || sb.y1 > p->Y || sb.x1 > p->X)
        return fail(PPP_ERR_INVALID_ARG, "score_box outside the volume");
    // the consensus tile must hold every base voxel the scored centres read
    const ppp_box &cb = p->cons_box;
    auto lo = [](int a, int r) { return a - r < 0 ? 0 : a - r; };
    auto hi = [](int a, int r, int n) { return a + r > n ? n : a + r; };
    if (lo(sb.z0, G.rz) < cb.z0 || lo(sb.y0, G.ry) < cb.y0 || lo(sb.x0, G.rx) < cb.x0 ||
        hi(sb.z1, G.rz, p->Z) > cb.z1 || hi(sb.y1, G.ry, p->Y) > cb.y1 || hi(sb.x1, G.rx, p->X) > cb.x1)
        return fail(PPP_ERR_INVALID_ARG, "cons_box does not cover score_box grown by the patch radius");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_rank(d_pred, pred_dtype, d_cons, d_overlap, d_score, sb, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_rank_patches");
}

static int rank_box(const ppp_box *score_box, const ppp_params *p, const ppp::Geo &G, ppp_box *sb) {
    *sb = ppp_box{0, 0, 0, p->Z, p->Y, p->X};
    if (score_box) *sb = *score_box;
    if (sb->z0 < 0 || sb->y0 < 0 || sb->x0 < 0 || sb->z1 > p->Z || sb->y1 > p->Y || sb->x1 > p->X ||
        sb->z1 <= sb->z0 || sb->y1 <= sb->y0 || sb->x1 <= sb->x0)
        return fail(PPP_ERR_INVALID_ARG, "score_box outside the volume");
    const ppp_box &cb = p->cons_box;
    auto lo = [](int a, int r) { return a - r < 0 ? 0 : a - r; };
    auto hi = [](int a, int r, int n) { return a + r > n ? n : a + r; };
    if (lo(sb->z0, G.rz) < cb.z0 || lo(sb->y0, G.ry) < cb.y0 || lo(sb->x0, G.rx) < cb.x0 ||
        hi(sb->z1, G.rz, p->Z) > cb.z1 || hi(sb->y1, G.ry, p->Y) > cb.y1 || hi(sb->x1, G.rx, p->X) > cb.x1)
        return fail(PPP_ERR_INVALID_ARG, "cons_box does not cover score_box grown by the patch radius");
    return PPP_OK;
}

int64_t ppp_rank_workspace_bytes(const ppp_box *score_box, const ppp_params *p) {
    ppp::Geo G;
    if (make_geo(p, &G) != PPP_OK) return -1;
    if (!ppp::rank_vm_supported(G)) return 0;
    // (rows in a ring: the workgroup-per-tile kernel only -- a caller planning a ring asks with ring_z set)
    if (G.ring && !ppp::rank_wg_supported(G)) return 0;
    ppp_box sb;
    if (rank_box(score_box, p, G, &sb) != PPP_OK) return -1;
    return (int64_t)ppp::rank_vm_workspace_bytes(sb, G);
}

int ppp_rank_patches_vm(const void *d_pred, int pred_dtype, const float *d_cons_vm,
                        const uint8_t *d_overlap, float *d_score, const ppp_box *score_box,
                        void *d_work, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_pred || !d_cons_vm || !d_score || !d_work) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.use_overlap && !d_overlap) return fail(PPP_ERR_INVALID_ARG, "use_overlap set but d_overlap is NULL");
    if (!ppp::rank_vm_supported(G))
        return fail(PPP_ERR_UNSUPPORTED, "ppp_rank_patches_vm: VOXEL_MAJOR layout, cubic patches of 3/5/7/9, "
                                         "no count_pos_neg (use ppp_rank_patches otherwise)");
    if (G.ring && !ppp::rank_wg_supported(G))
        return fail(PPP_ERR_UNSUPPORTED, "ring_z: only the workgroup-per-tile ranking kernel reads a ring of rows");
    ppp_box sb;
    PPP_TRY(rank_box(score_box, p, G, &sb));
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_rank_vm(d_pred, pred_dtype, d_cons_vm, d_overlap, d_score, sb, d_work, G,
                                       (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_rank_patches_vm");
}

int ppp_patch_graph(const void *d_pred, int pred_dtype, const float *d_cons,
                    const uint32_t *d_pairs, const uint32_t *d_order, uint64_t n_pairs,
                    float *d_aff, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (n_pairs == 0) return PPP_OK;
    if (n_pairs >= (1ull << 32)) return fail(PPP_ERR_UNSUPPORTED, "more than 2^32-1 pair rows");
    if (!d_pred || !d_cons || !d_pairs || !d_aff) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.ring) return fail(PPP_ERR_UNSUPPORTED, "ring_z: only ppp_patch_graph_by_patch reads a ring of rows");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_patch_graph(d_pred, pred_dtype, d_cons, d_pairs, d_order, n_pairs, d_aff, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_patch_graph");
}

int32_t ppp_patch_graph_by_patch_chunk(const ppp_params *p) {
    ppp::Geo G;
    if (make_geo(p, &G) != PPP_OK) return -1;
    return ppp::patch_graph_pa_chunk(G, false);
}

int32_t ppp_patch_graph_by_patch_chunk_small(const ppp_params *p) {
    ppp::Geo G;
    if (make_geo(p, &G) != PPP_OK) return -1;
    return ppp::patch_graph_pa_chunk(G, true);
}

int64_t ppp_patch_graph_lcg_words(int32_t dz, int32_t dy, int32_t dx, const ppp_params *p) {
    ppp::Geo G;
    if (make_geo(p, &G) != PPP_OK) return 0;
    return ppp::patch_graph_lcg_words(G, dz, dy, dx);
}

int ppp_patch_graph_lcg(const void *d_pred, int pred_dtype, const uint32_t *d_pairs,
                        const uint32_t *d_order, const int64_t *d_lcg_pos, int64_t n_lcg,
                        const int64_t *d_drop_off, uint64_t *d_drops, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (n_lcg <= 0) return PPP_OK;
    if (!d_pred || !d_pairs || !d_order || !d_lcg_pos || !d_drop_off || !d_drops)
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_patch_graph_lcg(d_pred, pred_dtype, d_pairs, d_order, (const long long *)d_lcg_pos,
                                               (long long)n_lcg, (const long long *)d_drop_off,
                                               (unsigned long long *)d_drops, G, (hipStream_t)stream);
    if (e == hipErrorNotSupported)
        return fail(PPP_ERR_UNSUPPORTED, "no per-patch kernel for this patch shape");
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_patch_graph_lcg");
}

int ppp_patch_graph_by_patch_chunked(const void *d_pred, int pred_dtype, const float *d_cons_vm,
                                     const uint32_t *d_pairs, const uint32_t *d_order,
                                     const int64_t *d_group_start, const int64_t *d_chunk_offsets,
                                     int32_t n_groups, int64_t n_blocks, int32_t chunk, float *d_aff,
                                     const ppp_params *p, void *stream) {
    return ppp_patch_graph_by_patch_lcg(d_pred, pred_dtype, d_cons_vm, d_pairs, d_order, d_group_start,
                                        d_chunk_offsets, n_groups, n_blocks, chunk, d_aff, nullptr, nullptr,
                                        p, stream);
}

int ppp_patch_graph_by_patch_lcg(const void *d_pred, int pred_dtype, const float *d_cons_vm,
                                 const uint32_t *d_pairs, const uint32_t *d_order,
                                 const int64_t *d_group_start, const int64_t *d_chunk_offsets,
                                 int32_t n_groups, int64_t n_blocks, int32_t chunk, float *d_aff,
                                 const int64_t *d_drop_off, const uint64_t *d_drops,
                                 const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (n_groups == 0) return PPP_OK;
    if (!d_pred || !d_cons_vm || !d_pairs || !d_order || !d_group_start || !d_chunk_offsets || !d_aff)
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.layout != PPP_CONS_VOXEL_MAJOR)
        return fail(PPP_ERR_INVALID_ARG, "ppp_patch_graph_by_patch reads the VOXEL_MAJOR layout");
    if ((d_drop_off == nullptr) != (d_drops == nullptr))
        return fail(PPP_ERR_INVALID_ARG, "d_drop_off and d_drops go together");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_patch_graph_pa(d_pred, pred_dtype, d_cons_vm, d_pairs, d_order,
                                              (const long long *)d_group_start,
                                              (const long long *)d_chunk_offsets, n_groups, n_blocks,
                                              chunk, d_aff, (const long long *)d_drop_off,
                                              (const unsigned long long *)d_drops, G, (hipStream_t)stream);
    if (e == hipErrorNotSupported)
        return fail(PPP_ERR_UNSUPPORTED, "no per-patch kernel for this patch shape / chunk size");
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_patch_graph_by_patch");
}

int ppp_patch_graph_by_patch(const void *d_pred, int pred_dtype, const float *d_cons_vm,
                             const uint32_t *d_pairs, const uint32_t *d_order,
                             const int64_t *d_group_start, const int64_t *d_chunk_offsets,
                             int32_t n_groups, int64_t n_blocks, float *d_aff,
                             const ppp_params *p, void *stream) {
    return ppp_patch_graph_by_patch_chunked(d_pred, pred_dtype, d_cons_vm, d_pairs, d_order, d_group_start,
                                            d_chunk_offsets, n_groups, n_blocks,
                                            ppp_patch_graph_by_patch_chunk(p), d_aff, p, stream);
}

size_t ppp_label_workspace_bytes(const ppp_params *p) {
    if (!p) return 0;
    return (size_t)24 * (size_t)p->Z * p->Y * p->X;
}

int ppp_label_begin(const uint32_t *d_nodes, uint64_t n_nodes, void *d_work, const ppp_params *p,
                    void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n_nodes == 0) return PPP_OK;
    if (!d_nodes || !d_work) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.V >= (1ll << 32)) return fail(PPP_ERR_UNSUPPORTED, "more than 2^32-1 voxels");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_label_begin(d_nodes, n_nodes, d_work, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_label_begin");
}

int ppp_label_add(const uint32_t *d_pairs, const float *d_aff, const int64_t *d_row_ids,
                  int64_t first_row_id, uint64_t n_pairs, void *d_work, const ppp_params *p,
                  void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n_pairs == 0) return PPP_OK;
    if (!d_pairs || !d_aff || !d_work) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_label_add(d_pairs, d_aff, (const long long *)d_row_ids,
                                         (long long)first_row_id, n_pairs, d_work, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_label_add");
}

int ppp_label_union_edges(const int64_t *d_a, const int64_t *d_b, uint64_t n, void *d_work,
                          const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n == 0) return PPP_OK;
    if (!d_a || !d_b || !d_work) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_label_union_edges((const long long *)d_a, (const long long *)d_b, n,
                                                 d_work, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_label_union_edges");
}

int ppp_label_finish(const uint32_t *d_nodes, uint64_t n_nodes, int64_t *d_node_key, void *d_work,
                     const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n_nodes == 0) return PPP_OK;
    if (!d_nodes || !d_node_key || !d_work) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_label_finish(d_nodes, n_nodes, (long long *)d_node_key, nullptr,
                                            d_work, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_label_finish");
}

int ppp_label_components(const uint32_t *d_pairs, const float *d_aff, uint64_t n_pairs,
                         const uint32_t *d_nodes, uint64_t n_nodes, uint32_t *d_node_key,
                         void *d_work, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n_nodes == 0) return PPP_OK;
    if (n_pairs >= (1ull << 31)) return fail(PPP_ERR_UNSUPPORTED, "more than 2^31-1 pair rows");
    if ((n_pairs && (!d_pairs || !d_aff)) || !d_nodes || !d_node_key || !d_work)
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_label(d_pairs, d_aff, n_pairs, d_nodes, n_nodes, d_node_key, d_work, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_label_components");
}

static int pair_box(const ppp_params *p, int32_t max_ps_dist, int *box, int *l1max) {
    if (max_ps_dist < 0) return fail(PPP_ERR_INVALID_ARG, "max_ps_dist < 0");
    box[0] = max_ps_dist * p->pz; box[1] = max_ps_dist * p->py; box[2] = max_ps_dist * p->px;
    *l1max = 2 * (p->pz + p->py + p->px);
    return PPP_OK;
}

int ppp_patch_pairs_count(const int32_t *d_sorted_zyx, int64_t n, int32_t max_ps_dist,
                          int64_t *d_counts, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n == 0) return PPP_OK;
    if (!d_sorted_zyx || !d_counts) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    int box[3], l1max;
    PPP_TRY(pair_box(p, max_ps_dist, box, &l1max));
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_pairs_count(d_sorted_zyx, n, box, l1max, d_counts, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_patch_pairs_count");
}

int ppp_patch_pairs_count_subset(const int32_t *d_sorted_zyx, int64_t n, int32_t max_ps_dist,
                                 const int64_t *d_subset, int64_t m, int64_t *d_counts,
                                 const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n == 0 || m == 0) return PPP_OK;
    if (!d_sorted_zyx || !d_counts || !d_subset) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    int box[3], l1max;
    PPP_TRY(pair_box(p, max_ps_dist, box, &l1max));
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_pairs_count(d_sorted_zyx, n, box, l1max, d_counts, (hipStream_t)stream, d_subset, m);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_patch_pairs_count_subset");
}

int ppp_patch_pairs_fill_subset(const int32_t *d_sorted_zyx, int64_t n, int32_t max_ps_dist,
                                const int64_t *d_subset, int64_t m, const int64_t *d_local_offsets,
                                const int64_t *d_global_offsets, int64_t n_local_rows,
                                int64_t n_rows_total, int32_t include_single, uint32_t *d_rows,
                                int64_t *d_row_ids, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n == 0 || m == 0) return PPP_OK;
    if (!d_sorted_zyx || !d_subset || !d_local_offsets || !d_global_offsets || !d_rows || !d_row_ids)
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    int box[3], l1max;
    PPP_TRY(pair_box(p, max_ps_dist, box, &l1max));
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_pairs_subset(d_sorted_zyx, n, box, l1max, d_subset, m, d_local_offsets,
                                            d_global_offsets, n_local_rows, n_rows_total, include_single,
                                            d_rows, (long long *)d_row_ids, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_patch_pairs_fill_subset");
}

int ppp_patch_pairs_fill(const int32_t *d_sorted_zyx, int64_t n, int32_t max_ps_dist,
                         const int64_t *d_offsets, int64_t n_pair_rows, int32_t include_single,
                         uint32_t *d_rows, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n == 0) return PPP_OK;
    if (!d_sorted_zyx || !d_offsets || !d_rows) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    int box[3], l1max;
    PPP_TRY(pair_box(p, max_ps_dist, box, &l1max));
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_pairs_fill(d_sorted_zyx, n, box, l1max, d_offsets, n_pair_rows, include_single, d_rows, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_patch_pairs_fill");
}

int ppp_pair_sort_keys(const uint32_t *d_rows, uint64_t n_rows, int64_t *d_keys,
                       const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n_rows == 0) return PPP_OK;
    if (!d_rows || !d_keys) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_pair_keys(d_rows, n_rows, d_keys, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_pair_sort_keys");
}

int ppp_pair_group_keys(const uint32_t *d_rows, uint64_t n_rows, int64_t *d_keys,
                        const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (n_rows == 0) return PPP_OK;
    if (!d_rows || !d_keys) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if ((int64_t)(4 * G.pz + 1) * (4 * G.py + 1) * (4 * G.px + 1) > (1 << 17))
        return fail(PPP_ERR_UNSUPPORTED, "patch shape too large for ppp_pair_group_keys");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_pair_group_keys(d_rows, n_rows, d_keys, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_pair_group_keys");
}

int ppp_paint_instances(const void *d_pred, int pred_dtype, const uint32_t *d_nodes,
                        const uint32_t *d_labels, uint64_t n_nodes, uint32_t *d_instances,
                        const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (n_nodes == 0) return PPP_OK;
    if (!d_pred || !d_nodes || !d_labels || !d_instances) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_paint(d_pred, pred_dtype, d_nodes, d_labels, n_nodes, d_instances, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_paint_instances");
}

// ---- the reference's NumPy-semantics stages (cuda=False) -----------------------------------------
int64_t ppp_np_vote_planes(const ppp_params *p) {
    ppp::Geo G;
    if (!p || make_geo(p, &G) != PPP_OK) return -1;
    return (int64_t)G.n_planes + 1;
}

int ppp_np_consensus(const void *d_pred, int pred_dtype, const uint8_t *d_foreground, int16_t *d_votes,
                     const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_pred || !d_foreground || !d_votes) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_np_consensus(d_pred, pred_dtype, d_foreground, d_votes, G, p->th, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_np_consensus");
}

int ppp_np_rank_patches(const void *d_pred, int pred_dtype, const uint8_t *d_foreground, const int16_t *d_votes,
                        int32_t *d_score, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_pred || !d_foreground || !d_votes || !d_score) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.C > 32768) return fail(PPP_ERR_UNSUPPORTED, "patch too large for ppp_np_rank_patches");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_np_rank(d_pred, pred_dtype, d_foreground, d_votes, d_score, G, p->th, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_np_rank_patches");
}

int ppp_np_patch_graph(const void *d_pred, int pred_dtype, const uint8_t *d_mask, const int16_t *d_votes,
                       const int32_t *d_rows, uint64_t n_rows, int64_t *d_weight, int32_t *d_count,
                       const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (n_rows == 0) return PPP_OK;
    if (!d_pred || !d_mask || !d_votes || !d_rows || !d_weight || !d_count) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.C > 16384) return fail(PPP_ERR_UNSUPPORTED, "patch too large for ppp_np_patch_graph");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_np_graph(d_pred, pred_dtype, d_mask, d_votes, d_rows, n_rows, (long long *)d_weight, d_count,
                                        G, p->th, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_np_patch_graph");
}

int ppp_paint_patch_rows(const void *d_rows, int rows_dtype, const uint32_t *d_nodes,
                         const uint32_t *d_labels, uint64_t n_nodes, uint32_t *d_instances,
                         const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(rows_dtype));
    if (n_nodes == 0) return PPP_OK;
    if (!d_rows || !d_nodes || !d_labels || !d_instances) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_paint_rows(d_rows, rows_dtype, d_nodes, d_labels, n_nodes, d_instances, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_paint_patch_rows");
}

int ppp_cons_to_reference(const float *d_cons_compact, float *d_cons_reference,
                          const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (!d_cons_compact || !d_cons_reference) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.BV != G.V) return fail(PPP_ERR_INVALID_ARG, "whole-volume consensus required");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_cons_to_reference(d_cons_compact, d_cons_reference, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_cons_to_reference");
}

int ppp_cons_to_voxel_major(const float *d_cons_compact, float *d_cons_voxel_major,
                            const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (!d_cons_compact || !d_cons_voxel_major) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.layout == PPP_CONS_REFERENCE) return fail(PPP_ERR_INVALID_ARG, "source must be a COMPACT consensus");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_cons_to_voxel_major(d_cons_compact, d_cons_voxel_major, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_cons_to_voxel_major");
}

int ppp_cons_planes_to_rows(const float *d_planes, const ppp_box *planes_box, float *d_rows, const ppp_params *p,
                            void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (!d_planes || !d_rows || !planes_box) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    if (G.layout != PPP_CONS_VOXEL_MAJOR || G.ring) return fail(PPP_ERR_INVALID_ARG, "params must describe VOXEL_MAJOR rows of a plain box");
    const ppp_box &b = p->cons_box, &q = *planes_box;
    if (q.z1 <= q.z0 || q.y1 <= q.y0 || q.x1 <= q.x0 || b.z0 < q.z0 || b.y0 < q.y0 || b.x0 < q.x0 || b.z1 > q.z1 ||
        b.y1 > q.y1 || b.x1 > q.x1)
        return fail(PPP_ERR_INVALID_ARG, "the planes' box must hold cons_box (the rows' box)");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_cons_planes_to_rows(d_planes, q, d_rows, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_cons_planes_to_rows");
}

int ppp_patch_bits(const void *d_pred, int pred_dtype, const uint32_t *d_centres, uint64_t n,
                   double thresh, uint32_t *d_bits, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (n == 0) return PPP_OK;
    if (!d_pred || !d_centres || !d_bits) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_patch_bits(d_pred, pred_dtype, d_centres, n, (float)thresh, d_bits, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_patch_bits");
}

int ppp_patch_bits_volume(const void *d_pred, int pred_dtype, double thresh, uint32_t *d_bits_vol,
                          const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_pred || !d_bits_vol) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_patch_bits_volume(d_pred, pred_dtype, (float)thresh, d_bits_vol, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_patch_bits_volume");
}

int ppp_synth_pred(const int32_t *d_labels, void *d_pred, int pred_dtype, uint32_t seed,
                   float hi, float lo, float noise, uint64_t voxel_offset, const ppp_params *p,
                   void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_labels || !d_pred) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_synth(d_labels, d_pred, pred_dtype, seed, hi, lo, noise, voxel_offset, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_synth_pred");
}

int ppp_synth_pred_box(const int32_t *d_labels, const int32_t *label_box, void *d_pred, int pred_dtype,
                       uint32_t seed, float hi, float lo, float noise, const int32_t *global_dims,
                       const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_labels || !d_pred || !label_box || !global_dims) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    const int lo3[3] = {G.oz, G.oy, G.ox}, ext[3] = {G.Z, G.Y, G.X}, rad[3] = {G.rz, G.ry, G.rx};
    for (int a = 0; a < 3; ++a) {
        const int need_lo = lo3[a] - rad[a] < 0 ? 0 : lo3[a] - rad[a];
        const int need_hi = lo3[a] + ext[a] + rad[a] > global_dims[a] ? global_dims[a] : lo3[a] + ext[a] + rad[a];
        if (lo3[a] < 0 || lo3[a] + ext[a] > global_dims[a] || label_box[a] > need_lo || label_box[3 + a] < need_hi)
            return fail(PPP_ERR_INVALID_ARG, "label box must hold the prediction box grown by the patch radius");
    }
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_synth_box(d_labels, label_box, d_pred, pred_dtype, seed, hi, lo, noise, global_dims,
                                         G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_synth_pred_box");
}

int ppp_counter_calibration(const void *d_src, int src_dtype, int64_t n_read, float *d_dst, int64_t n_write,
                            void *stream) {
    PPP_TRY(check_dtype(src_dtype));
    if ((n_read > 0 && !d_src) || !d_dst || n_read < 0 || n_write < 0) return fail(PPP_ERR_INVALID_ARG, "bad argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_counter_calibration(d_src, src_dtype, n_read, d_dst, n_write, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_counter_calibration");
}

int ppp_pred_check(const void *d_pred, int pred_dtype, int64_t n_values, int32_t *d_unclean, const ppp_params *p,
                   void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (!d_unclean || n_values < 0 || (n_values > 0 && !d_pred)) return fail(PPP_ERR_INVALID_ARG, "bad argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_pred_check(d_pred, pred_dtype, n_values, G, d_unclean, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_pred_check");
}

int ppp_decode_tail(const float *d_x, int64_t n, int32_t fmaps, int32_t side, const float *d_w1, float b1,
                    const float *d_w2, float b2, const float *d_w3, float b3, const int64_t *d_dst,
                    void *d_pred, int pred_dtype, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    PPP_TRY(check_dtype(pred_dtype));
    if (n <= 0) return PPP_OK;
    if (!d_x || !d_w1 || !d_w2 || !d_w3 || !d_dst || !d_pred) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    hipError_t e = ppp::launch_decode_tail(d_x, n, fmaps, side, d_w1, b1, d_w2, b2, d_w3, b3,
                                           (const long long *)d_dst, d_pred, pred_dtype, G, (hipStream_t)stream);
    if (e == hipErrorNotSupported)
        return fail(PPP_ERR_UNSUPPORTED, "ppp_decode_tail serves the shipped decoder tail only: 64 feature maps "
                                         "at 4^3, 3^3 kernels, 7^3 patches");
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_decode_tail");
}

int64_t ppp_cover_workspace_bytes(int64_t n, const ppp_params *p) {
    ppp::Geo G;
    if (make_geo(p, &G) != PPP_OK) return -1;
    return (int64_t)ppp::cover_workspace_bytes(n < 0 ? 0 : n, G);
}

static int cover_pass_impl(uint8_t *d_mask, const uint32_t *d_bits, int64_t first_voxel, const int64_t *d_lin,
                           int64_t n, int32_t pix_th, int32_t *d_state, int32_t *d_cleared, void *d_work,
                           const ppp_params *p, void *stream, int32_t *rounds, const char *who) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (rounds) *rounds = 0;
    if (n <= 0) return PPP_OK;
    if (n > 0x7F000000LL) return fail(PPP_ERR_INVALID_ARG, "too many ranked patches for one cover pass");
    if (G.px > 32) return fail(PPP_ERR_UNSUPPORTED, "%s needs patch rows of at most 32 voxels", who);
    if (!d_mask || !d_bits || !d_lin || !d_state || !d_cleared || !d_work)
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    int r = 0;
    hipError_t e = ppp::run_cover_pass(d_mask, d_bits, first_voxel, (const long long *)d_lin, n, pix_th, d_state,
                                       d_cleared, d_work, G, (hipStream_t)stream, &r);
    if (rounds) *rounds = r;
    return e == hipSuccess ? PPP_OK : hip_fail(e, who);
}

int ppp_cover_pass(uint8_t *d_mask, const uint32_t *d_bits, const int64_t *d_lin, int64_t n,
                   int32_t pix_th, int32_t *d_state, int32_t *d_cleared, void *d_work,
                   const ppp_params *p, void *stream, int32_t *rounds) {
    return cover_pass_impl(d_mask, d_bits, -1, d_lin, n, pix_th, d_state, d_cleared, d_work, p, stream, rounds,
                           "ppp_cover_pass");
}

int ppp_cover_pass_voxel_bits(uint8_t *d_mask, const uint32_t *d_bits_by_voxel, int64_t first_voxel,
                              const int64_t *d_lin, int64_t n, int32_t pix_th, int32_t *d_state,
                              int32_t *d_cleared, void *d_work, const ppp_params *p, void *stream,
                              int32_t *rounds) {
    if (first_voxel < 0) return fail(PPP_ERR_INVALID_ARG, "first_voxel must be >= 0");
    return cover_pass_impl(d_mask, d_bits_by_voxel, first_voxel, d_lin, n, pix_th, d_state, d_cleared, d_work, p,
                           stream, rounds, "ppp_cover_pass_voxel_bits");
}

int64_t ppp_rank_order_workspace_bytes(const ppp_params *p) {
    ppp::Geo G;
    if (make_geo(p, &G) != PPP_OK) return -1;
    return (int64_t)ppp::rank_order_workspace_bytes(G);
}

int ppp_rank_order(const float *d_score, const uint8_t *d_foreground, int64_t *d_lin,
                   float *d_rank_score, int64_t *count, void *d_work, const ppp_params *p,
                   void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (!d_score || !d_foreground || !d_lin || !count || !d_work)
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    long long n = 0;
    hipError_t e = ppp::run_rank_order(d_score, d_foreground, (long long *)d_lin, d_rank_score, &n, d_work,
                                       G, (hipStream_t)stream);
    *count = n;
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_rank_order");
}

int64_t ppp_mws_edges_workspace_bytes(int64_t n_rows, int64_t n_nodes, const ppp_params *p) {
    ppp::Geo G;
    if (make_geo(p, &G) != PPP_OK) return -1;
    return (int64_t)ppp::mws_edges_workspace_bytes(n_rows, n_nodes, G);
}

int ppp_mws_edges(const uint32_t *d_pairs, const float *d_aff, int64_t n_rows, const uint32_t *d_nodes,
                  int64_t n_nodes, int32_t *d_eu, int32_t *d_ev, int64_t *n_edges, void *d_work,
                  const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (!n_edges) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    *n_edges = 0;
    if (n_rows <= 0) return PPP_OK;
    if (!d_pairs || !d_aff || !d_nodes || !d_eu || !d_ev || !d_work)
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    long long ne = 0;
    hipError_t e = ppp::run_mws_edges(d_pairs, d_aff, n_rows, d_nodes, n_nodes, d_eu, d_ev, &ne, d_work, G,
                                      (hipStream_t)stream);
    *n_edges = ne;
    if (e == hipErrorInvalidValue)
        return fail(PPP_ERR_INVALID_ARG, "ppp_mws_edges: too many rows, or a row names a voxel that is not a node");
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_mws_edges");
}

int64_t ppp_thin_workspace_bytes(int64_t n, const ppp_params *p) {
    ppp::Geo G;
    if (make_geo(p, &G) != PPP_OK) return -1;
    return (int64_t)ppp::thin_workspace_bytes(n < 0 ? 0 : n, G);
}

int ppp_thin_cover(const uint8_t *d_mask, const uint32_t *d_bits, const int64_t *d_lin, int64_t n,
                   uint8_t *d_keep, void *d_work, const ppp_params *p, void *stream,
                   int32_t *rounds) {
    ppp::Geo G;
    PPP_TRY(make_geo(p, &G));
    if (rounds) *rounds = 0;
    if (n <= 0) return PPP_OK;
    if (n > 0x7F000000LL) return fail(PPP_ERR_INVALID_ARG, "too many selected patches for ppp_thin_cover");
    if (G.px > 32) return fail(PPP_ERR_UNSUPPORTED, "ppp_thin_cover needs patch rows of at most 32 voxels");
    if (!d_mask || !d_bits || !d_lin || !d_keep || !d_work)
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    PPP_TRY(need_device());
    int r = 0;
    hipError_t e = ppp::run_thin_cover(d_mask, d_bits, (const long long *)d_lin, n, d_keep, d_work, G,
                                       (hipStream_t)stream, &r);
    if (rounds) *rounds = r;
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_thin_cover");
}

/* ---- sharded cover: the rounds of ppp_cover_pass one step at a time on a rank's z-range ---- */
static int cover_geo(const ppp_params *p, ppp::Geo *G) {
    PPP_TRY(make_geo(p, G));
    if (G->px > 32) return fail(PPP_ERR_UNSUPPORTED, "the cover needs patch rows of at most 32 voxels");
    return need_device();
}

int ppp_cover_open(const uint8_t *d_mask, const int64_t *d_lin, const int32_t *d_rank_id, int64_t n,
                   const int32_t *d_state, int32_t *d_cleared, void *d_work, const ppp_params *p,
                   void *stream) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_mask || !d_work || (n > 0 && (!d_lin || !d_rank_id || !d_state || !d_cleared)))
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    hipError_t e = ppp::cover_open(d_mask, (const long long *)d_lin, d_rank_id, n, d_state, d_cleared,
                                   d_work, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_cover_open");
}

int ppp_cover_step(int32_t what, const uint32_t *d_bits, int32_t pix_th, int32_t *d_state,
                   int32_t *d_cleared, void *d_work, int32_t global_z, const ppp_params *p,
                   void *stream) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_work) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    hipError_t e;
    if (what == PPP_COVER_COUNT) e = ppp::cover_step_count(d_bits, pix_th, d_state, d_work, G, (hipStream_t)stream);
    else if (what == PPP_COVER_FILTER) e = ppp::cover_step_filter(d_work, G, (hipStream_t)stream);
    else if (what == PPP_COVER_SELECT) e = ppp::cover_step_select(d_bits, pix_th, d_state, d_cleared, d_work, global_z, G, (hipStream_t)stream);
    else return fail(PPP_ERR_INVALID_ARG, "unknown cover step %d", what);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_cover_step");
}

int ppp_cover_alive(void *d_work, const ppp_params *p, void *stream, int32_t *alive) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_work || !alive) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    hipError_t e = ppp::cover_alive(d_work, G, alive, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_cover_alive");
}

int ppp_cover_close(uint8_t *d_mask, void *d_work, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_mask || !d_work) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    hipError_t e = ppp::cover_close(d_mask, d_work, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_cover_close");
}

int ppp_cover_zone(int32_t import, void *d_work, int32_t z_lo, int32_t z_hi, int32_t own_lo,
                   int32_t own_hi, int32_t *d_rank, uint8_t *d_mask, uint8_t *d_clean,
                   const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_work || z_lo < 0 || z_hi > G.Z || z_lo > z_hi || ((d_mask == nullptr) != (d_clean == nullptr)))
        return fail(PPP_ERR_INVALID_ARG, "bad zone");
    hipError_t e = import ? ppp::cover_zone_import(d_work, z_lo, z_hi, d_rank, d_mask, d_clean, G, (hipStream_t)stream)
                          : ppp::cover_zone_export(d_work, z_lo, z_hi, own_lo, own_hi, d_rank, d_mask, d_clean, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_cover_zone");
}

/* ---- sharded thinning: the rounds of ppp_thin_cover one step at a time on a rank's z-range ---- */
int64_t ppp_thin_shard_workspace_bytes(const ppp_params *p) {
    ppp::Geo G;
    if (make_geo(p, &G) != PPP_OK) return -1;
    return (int64_t)ppp::thin_shard_workspace_bytes(G);
}

int ppp_thin_open(const uint8_t *d_mask, const int64_t *d_lin, const int32_t *d_index, int64_t n, int32_t *d_state,
                  int32_t *d_count, int32_t *d_cleared, void *d_work, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_mask || !d_work || (n > 0 && (!d_lin || !d_index || !d_state || !d_count || !d_cleared)))
        return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    hipError_t e = ppp::thin_open(d_mask, (const long long *)d_lin, d_index, n, d_state, d_count, d_cleared, d_work, G,
                                  (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_thin_open");
}

int ppp_thin_step(int32_t what, const uint32_t *d_bits, int32_t *d_state, int32_t *d_count, int32_t *d_cleared,
                  void *d_work, int32_t global_z, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_work) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    hipError_t e;
    if (what == PPP_COVER_COUNT) e = ppp::thin_step_count(d_bits, d_state, d_work, G, (hipStream_t)stream);
    else if (what == PPP_COVER_FILTER) e = ppp::thin_step_filter(d_work, G, (hipStream_t)stream);
    else if (what == PPP_COVER_SELECT) e = ppp::thin_step_select(d_bits, d_state, d_count, d_cleared, d_work, global_z, G, (hipStream_t)stream);
    else return fail(PPP_ERR_INVALID_ARG, "unknown thinning step %d", what);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_thin_step");
}

int ppp_thin_alive(void *d_work, const ppp_params *p, void *stream, int32_t *alive) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_work || !alive) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    hipError_t e = ppp::thin_alive(d_work, G, alive, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_thin_alive");
}

int ppp_thin_close(uint8_t *d_mask, void *d_work, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_mask || !d_work) return fail(PPP_ERR_INVALID_ARG, "NULL pointer argument");
    hipError_t e = ppp::thin_close(d_mask, d_work, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_thin_close");
}

int ppp_thin_zone(int32_t import, void *d_work, int32_t z_lo, int32_t z_hi, int32_t own_lo, int32_t own_hi,
                  int64_t *d_key, uint8_t *d_mask, uint8_t *d_clean, const ppp_params *p, void *stream) {
    ppp::Geo G;
    PPP_TRY(cover_geo(p, &G));
    if (!d_work || z_lo < 0 || z_hi > G.Z || z_lo > z_hi || ((d_mask == nullptr) != (d_clean == nullptr)))
        return fail(PPP_ERR_INVALID_ARG, "bad zone");
    hipError_t e = import ? ppp::thin_zone_import(d_work, z_lo, z_hi, (const long long *)d_key, d_mask, d_clean, G, (hipStream_t)stream)
                          : ppp::thin_zone_export(d_work, z_lo, z_hi, own_lo, own_hi, (long long *)d_key, d_mask, d_clean, G, (hipStream_t)stream);
    return e == hipSuccess ? PPP_OK : hip_fail(e, "ppp_thin_zone");
}

}  // extern "C"
