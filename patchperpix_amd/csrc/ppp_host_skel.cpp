// ppp_host_skel.cpp -- 3-d thinning of the foreground mask (host).
//
// Reference: vote_instances.py:219-224 and stitch_patch_graph.py:756-759 call
// skimage.morphology.skeletonize_3d (scikit-image, unpinned and absent from this image) when
// `skeletonize_foreground` is set -- the shipped flylight configuration does (default.toml:164):
// the cover mask (whole-volume mode) or the bounding box (blockwise mode) is that of the
// skeleton.  skeletonize_3d is the thinning of Lee, Kashyap and Chu, "Building skeleton models
// via 3-D medial surface / axis thinning algorithms" (CVGIP 1994), in the formulation that ITK
// and Fiji's Skeletonize3D popularised.  This file restates that published algorithm:
//
//   repeat until a pass over all six directions deletes nothing:
//     for every border direction (order: 4, 3, 2, 1, 5, 6 = -y, +y, +x, -x, +z, -z neighbour empty)
//       candidates = foreground voxels that are border voxels for the direction, are not arc
//                    end points (exactly one 26-neighbour), whose deletion keeps the Euler
//                    characteristic of their 3x3x3 neighbourhood, and that are simple points
//                    (the 26-neighbours stay ONE 26-connected component without the voxel);
//       the candidates are re-checked one at a time in raster order against the CURRENT image
//       (sequential re-check: keeps connectivity when neighbours go in the same sub-iteration)
//       and deleted when all three conditions still hold.
//
// PARITY UNPINNED: with scikit-image absent there is no output of the reference's dependency to
// compare with; the tests check the properties the reference relies on (a subset of the mask,
// same number of connected components, thin, idempotent, analytic cases).
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

// Euler characteristic change of the union of closed unit cubes when the centre cube of a
// 3x3x3 neighbourhood nb (0 / 1, index (dz*3 + dy)*3 + dx) is removed: cells of the centre cube
// that no other foreground cube shares.  chi = V - E + F - C.
int euler_delta(const uint8_t *nb) {
    // count the cells of the centre cube that it owns exclusively
    int faces = 0, edges = 0, verts = 0;
    // faces: shared with the face neighbour
    static const int F[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    for (int f = 0; f < 6; ++f)
        if (!nb[((1 + F[f][0]) * 3 + (1 + F[f][1])) * 3 + (1 + F[f][2])]) ++faces;
    // edges: an edge of the centre cube is shared by the 3 other cubes around it
    for (int ax = 0; ax < 3; ++ax)                  // edge direction
        for (int s1 = -1; s1 <= 1; s1 += 2)
            for (int s2 = -1; s2 <= 1; s2 += 2) {
                int o1[3] = {0, 0, 0}, o2[3] = {0, 0, 0};
                o1[(ax + 1) % 3] = s1;
                o2[(ax + 2) % 3] = s2;
                const bool a = nb[((1 + o1[0]) * 3 + (1 + o1[1])) * 3 + (1 + o1[2])];
                const bool b = nb[((1 + o2[0]) * 3 + (1 + o2[1])) * 3 + (1 + o2[2])];
                const bool c = nb[((1 + o1[0] + o2[0]) * 3 + (1 + o1[1] + o2[1])) * 3 + (1 + o1[2] + o2[2])];
                if (!a && !b && !c) ++edges;
            }
    // vertices: shared by the 7 other cubes of the octant
    for (int sz = -1; sz <= 1; sz += 2)
        for (int sy = -1; sy <= 1; sy += 2)
            for (int sx = -1; sx <= 1; sx += 2) {
                bool any = false;
                for (int m = 1; m < 8; ++m) {
                    const int dz = (m & 4) ? sz : 0, dy = (m & 2) ? sy : 0, dx = (m & 1) ? sx : 0;
                    any |= nb[((1 + dz) * 3 + (1 + dy)) * 3 + (1 + dx)] != 0;
                }
                if (!any) ++verts;
            }
    // chi(with) - chi(without) = verts - edges + faces - 1
    return verts - edges + faces - 1;
}

// the 26-neighbours of the centre form exactly one 26-connected component (centre excluded)
bool one_component(const uint8_t *nb) {
    int first = -1, count = 0;
    for (int i = 0; i < 27; ++i)
        if (i != 13 && nb[i]) { if (first < 0) first = i; ++count; }
    if (count == 0) return false;
    uint8_t seen[27] = {0};
    int stack[27], sp = 0, reached = 1;
    stack[sp++] = first;
    seen[first] = 1;
    while (sp) {
        const int i = stack[--sp];
        const int z = i / 9, y = (i / 3) % 3, x = i % 3;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int zz = z + dz, yy = y + dy, xx = x + dx;
                    if (zz < 0 || zz > 2 || yy < 0 || yy > 2 || xx < 0 || xx > 2) continue;
                    const int j = (zz * 3 + yy) * 3 + xx;
                    if (j == 13 || !nb[j] || seen[j]) continue;
                    seen[j] = 1;
                    stack[sp++] = j;
                    ++reached;
                }
    }
    return reached == count;
}

}  // namespace

extern "C" {

// mask: uint8 (Z, Y, X), 0 / non-zero; out: uint8 (Z, Y, X) 0 / 1, the thinned mask.
// Returns the number of voxels kept, or -1 on bad arguments.
int64_t ppp_host_skeletonize_3d(const uint8_t *mask, const int32_t *vol, uint8_t *out) {
    if (!mask || !vol || !out) return -1;
    const int Z = vol[0], Y = vol[1], X = vol[2];
    if (Z <= 0 || Y <= 0 || X <= 0) return -1;
    // padded working image (one empty voxel all around)
    const int pZ = Z + 2, pY = Y + 2, pX = X + 2;
    std::vector<uint8_t> img((size_t)pZ * pY * pX, 0);
    auto at = [&](int z, int y, int x) -> uint8_t & { return img[((size_t)z * pY + y) * pX + x]; };
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) at(z + 1, y + 1, x + 1) = mask[((size_t)z * Y + y) * X + x] ? 1 : 0;
    auto neighbourhood = [&](int z, int y, int x, uint8_t *nb) {
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) nb[((dz + 1) * 3 + (dy + 1)) * 3 + (dx + 1)] = at(z + dz, y + dy, x + dx);
    };
    auto removable = [&](const uint8_t *nb) -> bool {
        int n = 0;
        for (int i = 0; i < 27; ++i) n += nb[i];
        if (n == 2) return false;                      // arc end point (itself + one neighbour)
        if (euler_delta(nb) != 0) return false;
        return one_component(nb);
    };
    // border directions in the order 4, 3, 2, 1, 5, 6 of the published implementation:
    // the neighbour that must be empty, as (dz, dy, dx)
    static const int DIRS[6][3] = {{0, -1, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, -1}, {1, 0, 0}, {-1, 0, 0}};
    const int n_dirs = Z > 1 ? 6 : 4;                  // a single slice has no z borders to peel
    int unchanged = 0;
    std::vector<int> cand;
    uint8_t nb[27];
    while (unchanged < n_dirs) {
        unchanged = 0;
        for (int d = 0; d < n_dirs; ++d) {
            cand.clear();
            for (int z = 1; z <= Z; ++z)
                for (int y = 1; y <= Y; ++y)
                    for (int x = 1; x <= X; ++x) {
                        if (!at(z, y, x)) continue;
                        if (at(z + DIRS[d][0], y + DIRS[d][1], x + DIRS[d][2])) continue;   // not a border voxel
                        neighbourhood(z, y, x, nb);
                        if (removable(nb)) { cand.push_back(z); cand.push_back(y); cand.push_back(x); }
                    }
            bool changed = false;
            // sequential re-check against the CURRENT image.  All three conditions again: with
            // the simple-point test alone a plate two voxels thick unravels -- once one of its
            // rows is gone the other row's voxels have become arc end points, each still
            // "simple", and the arc is eaten from its end down to nothing.
            for (size_t i = 0; i < cand.size(); i += 3) {
                const int z = cand[i], y = cand[i + 1], x = cand[i + 2];
                neighbourhood(z, y, x, nb);
                if (removable(nb)) { at(z, y, x) = 0; changed = true; }
            }
            if (!changed) ++unchanged;
        }
    }
    int64_t kept = 0;
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                const uint8_t v = at(z + 1, y + 1, x + 1);
                out[((size_t)z * Y + y) * X + x] = v;
                kept += v;
            }
    return kept;
}

}  // extern "C"
