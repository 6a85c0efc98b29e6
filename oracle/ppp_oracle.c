/*
 * ppp_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the arithmetic of the four device kernels on the
 * PatchPerPix vote_instances path, with every shape / threshold / build flag turned
 * into a run-time parameter.  It exists so that the HIP kernels in
 * patchperpix_amd/csrc/ can be checked bit-for-bit on machines where the reference
 * tree is absent.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; the product path never does.
 *
 * PINNING: this restatement is checked against golden vectors produced by running
 * the reference's own kernels + Python stage functions in the development
 * container (tests/golden/gen_golden.py, tests/test_oracle_golden.py): consensus,
 * scores and patch affinities are bit-identical on every committed case.
 *
 * Serialisation: the reference kernels accumulate with float atomicAdd, so any
 * thread order is a legal execution.  The oracle runs the "threads" (one per voxel)
 * in raster order (z, y, x) -- the canonical order of the golden vectors.
 *
 * Reference files restated (all under PatchPerPix/vote_instances/):
 *   cuda/fillConsensusArray.cu:5-218      -> ppp_oracle_fill_consensus
 *   cuda/normConsensusArray.cu:5-43       -> ppp_oracle_norm_consensus
 *   cuda/rankPatches.cu:1-161             -> ppp_oracle_rank_patches
 *   cuda/computePatchGraph.cu:3-136       -> ppp_oracle_patch_graph
 *   utilVoteInstances.py:340-449          -> the parameter struct (macro
 *                                            substitution + -D build flags)
 * Consensus layout is the reference's [NSZ][NSY][NSX][Z][Y][X] (consensus_array.py:
 * 99-106), NS = 2*p per axis (NSZ = 1 for 2-d data, vote_instances.py:249-253).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { PPP_BG_INV_TH = 0, PPP_BG_HALF_TH = 1, PPP_BG_LESS_THAN_TH = 2 };
enum { PPP_VAL_COUNT = 0, PPP_VAL_PROB_PRODUCT = 1, PPP_VAL_NORM_PROB_PRODUCT = 2 };

typedef struct {
    int32_t Z, Y, X;       /* DATAZSIZE, DATAYSIZE, DATAXSIZE                      */
    int32_t pz, py, px;    /* PSZ, PSY, PSX                                        */
    int32_t nsz, nsy, nsx; /* NSZ, NSY, NSX                                        */
    double th;             /* TH: Python float repr pasted as a C double literal   */
    double thi;            /* THI (utilVoteInstances.py:361-365)                   */
    int32_t bg_rule;       /* -DUSE_INV_TH / -DUSE_HALF_TH / -DUSE_LESS_THAN_TH    */
    int32_t value_rule;    /* (none) / -DPROB_PRODUCT / -DNORM_PROB_PRODUCT        */
    int32_t use_overlap;   /* -DOVERLAP                                            */
    int32_t norm_rank;     /* -DNORM_PATCH_RANK                                    */
    int32_t count_pos_neg; /* -DCOUNT_POS_NEG                                      */
    int32_t norm_aff;      /* -DNORM_PATCH_AFFINITY                                */
    int32_t oz, oy, ox;    /* global coordinate of local voxel (0,0,0): the reference's LCG
                              seed uses absolute coordinates (tests of slab decompositions) */
} ppp_oracle_params;

/* ---- small helpers ---------------------------------------------------------- */
typedef struct {
    const ppp_oracle_params *P;
    size_t V;       /* voxels per channel / per consensus plane */
    int rz, ry, rx; /* patch radii */
    int C, mid;
} geom;

static geom make_geom(const ppp_oracle_params *P) {
    geom g;
    g.P = P;
    g.V = (size_t)P->Z * P->Y * P->X;
    g.rz = P->pz / 2;
    g.ry = P->py / 2;
    g.rx = P->px / 2;
    g.C = P->pz * P->py * P->px;
    g.mid = g.C / 2;
    return g;
}
static inline size_t vox(const geom *g, int z, int y, int x) {
    return ((size_t)z * g->P->Y + y) * g->P->X + x;
}
static inline size_t plane(const geom *g, int zo, int yo, int xo) {
    return (((size_t)zo * g->P->nsy + yo) * g->P->nsx + xo) * g->V;
}
static inline int interior(const geom *g, int z, int y, int x) {
    const ppp_oracle_params *P = g->P;
    return x >= g->rx && x < P->X - g->rx && y >= g->ry && y < P->Y - g->ry &&
           z >= g->rz && z < P->Z - g->rz;
}
/* background test on the second pixel of a pair; all compares in double because
 * the thresholds are double literals in the templated source */
static inline int is_bg(const ppp_oracle_params *P, float v2) {
    switch (P->bg_rule) {
    case PPP_BG_INV_TH: return (double)v2 < P->thi;
    case PPP_BG_HALF_TH: return (double)v2 < P->th / 2;
    default: return (double)v2 < P->th;
    }
}
static inline float vote_value(const ppp_oracle_params *P, float a, float b_or_1mb) {
    /* a * b is a float product; normalisation is evaluated in double and rounded
     * to float on assignment (fillConsensusArray.cu:105,128) */
    if (P->value_rule == PPP_VAL_NORM_PROB_PRODUCT)
        return (float)(((double)(a * b_or_1mb) - P->th * P->th) / (1.0 - P->th * P->th));
    if (P->value_rule == PPP_VAL_PROB_PRODUCT) return a * b_or_1mb;
    return 1.0f;
}

/* ---- S1: consensus fill (fillConsensusArray.cu:5-175, one call per "thread") --- */
static void fill_one(const geom *g, const float *pred, const uint8_t *ov, float *cons,
                     float *cnt, int cz, int cy, int cx) {
    const ppp_oracle_params *P = g->P;
    const size_t V = g->V;
    const size_t c = vox(g, cz, cy, cx);
    if (!interior(g, cz, cy, cx)) return;
    if ((double)pred[(size_t)g->mid * V + c] <= P->th) return;

    for (int a = 0, z1o = 0; z1o < P->pz; z1o++)
        for (int y1o = 0; y1o < P->py; y1o++)
            for (int x1o = 0; x1o < P->px; x1o++, a++) {
                const float v1 = pred[(size_t)a * V + c];
                if ((double)v1 <= P->th) continue;
                const int z1 = cz + z1o - g->rz, y1 = cy + y1o - g->ry, x1 = cx + x1o - g->rx;
                const size_t u1 = vox(g, z1, y1, x1);
                if ((double)pred[(size_t)g->mid * V + u1] <= P->th) continue;
                if (P->use_overlap && ov[u1] != 0) continue;

                for (int b = 0, z2o = 0; z2o < P->pz; z2o++)
                    for (int y2o = 0; y2o < P->py; y2o++)
                        for (int x2o = 0; x2o < P->px; x2o++, b++) {
                            if (a == b) continue;
                            const int z2 = cz + z2o - g->rz, y2 = cy + y2o - g->ry,
                                      x2 = cx + x2o - g->rx;
                            const size_t u2 = vox(g, z2, y2, x2);
                            if ((double)pred[(size_t)g->mid * V + u2] <= P->th) continue;
                            if (P->use_overlap && ov[u2] != 0) continue;
                            const float v2 = pred[(size_t)b * V + c];

                            if ((double)v2 > P->th) {
                                if (b <= a) continue; /* count each fg/fg pair once */
                                const size_t k = plane(g, z2o - z1o + P->pz - 1,
                                                       y2o - y1o + P->py - 1,
                                                       x2o - x1o + P->px - 1) + u1;
                                if (cnt) cnt[k] = cnt[k] + 1.0f;
                                if (cons) cons[k] = cons[k] + vote_value(P, v1, v2);
                            } else if (is_bg(P, v2)) {
                                size_t k;
                                if (b <= a) /* key = (later - earlier offset, earlier voxel) */
                                    k = plane(g, z1o - z2o + P->pz - 1, y1o - y2o + P->py - 1,
                                              x1o - x2o + P->px - 1) + u2;
                                else
                                    k = plane(g, z2o - z1o + P->pz - 1, y2o - y1o + P->py - 1,
                                              x2o - x1o + P->px - 1) + u1;
                                if (cnt) cnt[k] = cnt[k] + 1.0f;
                                if (cons) cons[k] = cons[k] + (-vote_value(P, v1, 1 - v2));
                            }
                        }
            }
}

/* cons and/or cnt may be NULL (the reference's default / -DOUTPUT_CNT / -DOUTPUT_BOTH
 * variants); both are accumulated INTO (caller zero-initialises). */
void ppp_oracle_fill_consensus(const float *pred, const uint8_t *overlap, float *cons,
                               float *cnt, const ppp_oracle_params *P) {
    geom g = make_geom(P);
    for (int z = 0; z < P->Z; z++)
        for (int y = 0; y < P->Y; y++)
            for (int x = 0; x < P->X; x++) fill_one(&g, pred, overlap, cons, cnt, z, y, x);
}

/* ---- S1 in gather form: one consensus entry at a time ------------------------------
 * The same numbers as ppp_oracle_fill_consensus (+ count pass + normalisation), organised so
 * that many cores can share the work WITHOUT atomics and without changing a bit: every key
 * (offset q, base voxel u) receives at most one vote per patch centre,
 * so its value is the float sum of those votes in raster order of the centres -- which is what
 * the serial scatter loop produces.  Threads take whole offset planes.  For a centre c that
 * holds both voxels, with pe = pred[pixel of u][c] (earlier pixel), pl = pred[pixel of u+q][c]:
 *   pe > TH and pl > TH      -> +value(pe, pl)        (fillConsensusArray.cu:94-113, b > a)
 *   pe > TH and pl is bg     -> -value(pe, 1 - pl)    (:114-166, b > a: key at the first pixel)
 *   pl > TH and pe is bg     -> -value(pl, 1 - pe)    (:114-166, b <= a: key at the second pixel)
 * cons (reference layout) is fully written for the lexicographically positive planes; cnt may
 * be NULL.  normalise: cons /= cnt where cnt != 0 (normConsensusArray.cu:23). */
void ppp_oracle_fill_consensus_planes(const float *pred, const uint8_t *ov, float *cons, float *cnt,
                                      int normalise, const ppp_oracle_params *P) {
    geom g = make_geom(P);
    const size_t V = g.V;
    const int wy = 2 * P->py - 1, wx = 2 * P->px - 1;
    const long n_off = (long)P->pz * wy * wx;
    const float *mid = pred + (size_t)g.mid * V;
#pragma omp parallel for schedule(dynamic, 1)
    for (long o = 0; o < n_off; o++) {
        const int dz = (int)(o / (wy * wx)), dy = (int)((o / wx) % wy) - (P->py - 1),
                  dx = (int)(o % wx) - (P->px - 1);
        if (dz == 0 && (dy < 0 || (dy == 0 && dx <= 0))) continue; /* not lexicographically > 0 */
        const size_t pl_off = plane(&g, dz + P->pz - 1, dy + P->py - 1, dx + P->px - 1);
        const int q_lin = (dz * P->py + dy) * P->px + dx; /* channel of the later pixel - earlier */
        for (int uz = 0; uz + dz < P->Z; uz++)
            for (int uy = (dy < 0 ? -dy : 0); uy < P->Y - (dy > 0 ? dy : 0); uy++)
                for (int ux = (dx < 0 ? -dx : 0); ux < P->X - (dx > 0 ? dx : 0); ux++) {
                    const size_t u = vox(&g, uz, uy, ux), w = vox(&g, uz + dz, uy + dy, ux + dx);
                    if ((double)mid[u] <= P->th || (double)mid[w] <= P->th) continue;
                    if (P->use_overlap && (ov[u] != 0 || ov[w] != 0)) continue;
                    /* centres whose window holds u and u + q, inside the interior */
                    int z0 = uz + dz - g.rz, z1 = uz + g.rz;
                    int y0 = (dy > 0 ? uy + dy : uy) - g.ry, y1 = (dy > 0 ? uy : uy + dy) + g.ry;
                    int x0 = (dx > 0 ? ux + dx : ux) - g.rx, x1 = (dx > 0 ? ux : ux + dx) + g.rx;
                    if (z0 < g.rz) z0 = g.rz;
                    if (y0 < g.ry) y0 = g.ry;
                    if (x0 < g.rx) x0 = g.rx;
                    if (z1 > P->Z - g.rz - 1) z1 = P->Z - g.rz - 1;
                    if (y1 > P->Y - g.ry - 1) y1 = P->Y - g.ry - 1;
                    if (x1 > P->X - g.rx - 1) x1 = P->X - g.rx - 1;
                    float sum = 0.0f, n = 0.0f;
                    for (int cz = z0; cz <= z1; cz++)
                        for (int cy = y0; cy <= y1; cy++)
                            for (int cx = x0; cx <= x1; cx++) {
                                const size_t c = vox(&g, cz, cy, cx);
                                if ((double)mid[c] <= P->th) continue;
                                const int a = ((uz - cz + g.rz) * P->py + (uy - cy + g.ry)) * P->px +
                                              (ux - cx + g.rx);
                                const float pe = pred[(size_t)a * V + c];
                                const float pl = pred[(size_t)(a + q_lin) * V + c];
                                if ((double)pe > P->th) {
                                    if ((double)pl > P->th) {
                                        sum = sum + vote_value(P, pe, pl);
                                        n = n + 1.0f;
                                    } else if (is_bg(P, pl)) {
                                        sum = sum + (-vote_value(P, pe, 1 - pl));
                                        n = n + 1.0f;
                                    }
                                } else if ((double)pl > P->th && is_bg(P, pe)) {
                                    sum = sum + (-vote_value(P, pl, 1 - pe));
                                    n = n + 1.0f;
                                }
                            }
                    cons[pl_off + u] = (normalise && n != 0) ? sum / n : sum;
                    if (cnt) cnt[pl_off + u] = n;
                }
    }
}

/* ---- normalisation (normConsensusArray.cu:5-28) -------------------------------- */
void ppp_oracle_norm_consensus(const float *pred, float *cons, const float *cnt,
                               const ppp_oracle_params *P) {
    geom g = make_geom(P);
    const size_t n_planes = (size_t)P->nsz * P->nsy * P->nsx;
#pragma omp parallel for schedule(static)
    for (size_t v = 0; v < g.V; v++) {
        if ((double)pred[(size_t)g.mid * g.V + v] <= P->th) continue;
        for (size_t o = 0; o < n_planes; o++) {
            const size_t k = o * g.V + v;
            if (cnt[k] != 0) cons[k] = cons[k] / cnt[k];
        }
    }
}

/* ---- S2: patch ranking (rankPatches.cu:1-161) ---------------------------------- */
static inline float signed_unit(float v3) { return v3 != 0 ? copysignf(1.0f, v3) : -1.0f; }

void ppp_oracle_rank_patches(const float *pred, const float *cons, const uint8_t *overlap,
                             float *score, const ppp_oracle_params *P) {
    geom g = make_geom(P);
    const size_t V = g.V;
    /* every centre is independent (own accumulator, own output): threads over (z, y) lines */
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int cz = 0; cz < P->Z; cz++)
        for (int cy = 0; cy < P->Y; cy++)
            for (int cx = 0; cx < P->X; cx++) {
                const size_t c = vox(&g, cz, cy, cx);
                if (!interior(&g, cz, cy, cx)) {
                    score[c] = P->norm_rank ? -1.0f : -9999999.0f;
                    continue;
                }
                if ((double)pred[(size_t)g.mid * V + c] <= P->th) continue; /* keeps 0 */

                float acc = 0.0f;
                unsigned fg_cnt = 0;
                for (int a = 0, z1o = 0; z1o < P->pz; z1o++)
                    for (int y1o = 0; y1o < P->py; y1o++)
                        for (int x1o = 0; x1o < P->px; x1o++, a++) {
                            const float v1 = pred[(size_t)a * V + c];
                            if ((double)v1 <= P->th) continue;
                            const size_t u1 =
                                vox(&g, cz + z1o - g.rz, cy + y1o - g.ry, cx + x1o - g.rx);
                            if ((double)pred[(size_t)g.mid * V + u1] <= P->th) continue;
                            if (P->use_overlap && overlap[u1] != 0) continue;
                            for (int b = 0, z2o = 0; z2o < P->pz; z2o++)
                                for (int y2o = 0; y2o < P->py; y2o++)
                                    for (int x2o = 0; x2o < P->px; x2o++, b++) {
                                        if (a == b) continue;
                                        const size_t u2 = vox(&g, cz + z2o - g.rz,
                                                              cy + y2o - g.ry, cx + x2o - g.rx);
                                        if ((double)pred[(size_t)g.mid * V + u2] <= P->th)
                                            continue;
                                        if (P->use_overlap && overlap[u2] != 0) continue;
                                        const float v2 = pred[(size_t)b * V + c];
                                        if ((double)v2 > P->th) {
                                            if (b <= a) continue;
                                            const float v3 =
                                                cons[plane(&g, z2o - z1o + P->pz - 1,
                                                           y2o - y1o + P->py - 1,
                                                           x2o - x1o + P->px - 1) + u1];
                                            if (P->count_pos_neg) acc += signed_unit(v3);
                                            else acc += v3;
                                        } else if (is_bg(P, v2)) {
                                            float v3;
                                            if (b <= a)
                                                v3 = cons[plane(&g, z1o - z2o + P->pz - 1,
                                                                y1o - y2o + P->py - 1,
                                                                x1o - x2o + P->px - 1) + u2];
                                            else
                                                v3 = cons[plane(&g, z2o - z1o + P->pz - 1,
                                                                y2o - y1o + P->py - 1,
                                                                x2o - x1o + P->px - 1) + u1];
                                            if (P->count_pos_neg) {
                                                if (v3 != 0) acc -= copysignf(1.0f, v3);
                                                else acc -= 1;
                                            } else acc -= v3;
                                        }
                                        fg_cnt += 1; /* also for dead-zone pairs */
                                    }
                        }
                if (P->norm_rank) score[c] = acc / (float)(fg_cnt > 1u ? fg_cnt : 1u);
                else score[c] = acc;
            }
}

/* ---- S5: patch graph (computePatchGraph.cu:3-136) ------------------------------- */
void ppp_oracle_patch_graph(const float *pred, const float *cons, const uint32_t *pairs,
                            uint64_t n_pairs, float *aff, const ppp_oracle_params *P) {
    geom g = make_geom(P);
    const size_t V = g.V;
    /* every pair row is independent */
#pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t id = 0; id < n_pairs; id++) {
        const int az = (int)pairs[id * 6 + 0], ay = (int)pairs[id * 6 + 1],
                  ax = (int)pairs[id * 6 + 2];
        const int bz = (int)pairs[id * 6 + 3], by = (int)pairs[id * 6 + 4],
                  bx = (int)pairs[id * 6 + 5];
        uint32_t rnd = (uint32_t)(az + P->oz) * (uint32_t)(bz + P->oz) * (uint32_t)(ay + P->oy) *
                       (uint32_t)(by + P->oy) * (uint32_t)(ax + P->ox) * (uint32_t)(bx + P->ox);
        const size_t ca = vox(&g, az, ay, ax), cb = vox(&g, bz, by, bx);
        float acc = 0.0f;
        unsigned fg_cnt = 0;

        for (int a = 0, z1o = 0; z1o < P->pz; z1o++)
            for (int y1o = 0; y1o < P->py; y1o++)
                for (int x1o = 0; x1o < P->px; x1o++, a++) {
                    const int z1 = az + z1o - g.rz, y1 = ay + y1o - g.ry, x1 = ax + x1o - g.rx;
                    const size_t u1 = vox(&g, z1, y1, x1);
                    if ((double)pred[(size_t)g.mid * V + u1] <= P->th) continue;
                    if ((double)pred[(size_t)a * V + ca] <= P->th) continue;
                    for (int b = 0, z2o = 0; z2o < P->pz; z2o++)
                        for (int y2o = 0; y2o < P->py; y2o++)
                            for (int x2o = 0; x2o < P->px; x2o++, b++) {
                                const int z2 = bz + z2o - g.rz, y2 = by + y2o - g.ry,
                                          x2 = bx + x2o - g.rx;
                                const size_t u2 = vox(&g, z2, y2, x2);
                                if ((double)pred[(size_t)g.mid * V + u2] <= P->th) continue;
                                if ((double)pred[(size_t)b * V + cb] <= P->th) continue;

                                const int g1 = x1 + P->X * y1 + P->X * P->Y * z1;
                                const int g2 = x2 + P->X * y2 + P->X * P->Y * z2;
                                /* both pixels inside both patches: keep ~20 % (LCG) */
                                if (abs(x1 - bx) <= g.rx && abs(y1 - by) <= g.ry &&
                                    abs(z1 - bz) <= g.rz && abs(x2 - ax) <= g.rx &&
                                    abs(y2 - ay) <= g.ry && abs(z2 - az) <= g.rz) {
                                    rnd = rnd * 1103515245U;
                                    const float rnd_t = rnd / 4294967296.0f;
                                    if ((double)rnd_t > 0.2) continue;
                                }
                                int zo, yo, xo;
                                size_t base;
                                if (g1 <= g2) {
                                    zo = z2 - z1; yo = y2 - y1; xo = x2 - x1; base = u1;
                                } else {
                                    zo = z1 - z2; yo = y1 - y2; xo = x1 - x2; base = u2;
                                }
                                zo += P->pz - 1; yo += P->py - 1; xo += P->px - 1;
                                if (zo < 0 || zo >= 2 * P->pz || yo < 0 || yo >= 2 * P->py ||
                                    xo < 0 || xo >= 2 * P->px)
                                    continue;
                                acc += cons[plane(&g, zo, yo, xo) + base];
                                fg_cnt += 1;
                            }
                }
        if (P->norm_aff) aff[id] = acc / (float)(fg_cnt > 1u ? fg_cnt : 1u);
        else aff[id] = acc;
    }
}

/* ---- whole kernel chain for one volume, used as the timed CPU baseline ---------- */
/* Runs S1 (+count pass +normalise when requested) and S2 in the reference's launch
 * order (consensus_array.py:124-193, ranked_patches.py:57-67). */
void ppp_oracle_consensus_and_rank(const float *pred, const uint8_t *overlap, float *cons,
                                   float *cnt, float *score, int normalise,
                                   const ppp_oracle_params *P) {
    ppp_oracle_fill_consensus(pred, overlap, cons, NULL, P);
    if (normalise) {
        ppp_oracle_fill_consensus(pred, overlap, NULL, cnt, P);
        ppp_oracle_norm_consensus(pred, cons, cnt, P);
    }
    ppp_oracle_rank_patches(pred, cons, overlap, score, P);
}
