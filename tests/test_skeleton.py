"""`skeletonize_foreground` (vote_instances.py:219-224, stitch_patch_graph.py:756-759): the 3-d
thinning the reference takes from scikit-image (skeletonize_3d = Lee / Kashyap / Chu 1994).
scikit-image is absent here: PARITY UNPINNED; what is tested are the properties the reference relies
on -- the skeleton is a subset of the mask with the same topology (components, tunnels), it is thin,
thinning it again changes nothing -- and analytic cases."""
import numpy as np
import pytest
from scipy import ndimage

from patchperpix_amd import backend

S26 = np.ones((3, 3, 3))


def euler_characteristic(vox):
    """chi of the union of closed unit cubes (26-connectivity of the foreground)"""
    v = np.pad(np.asarray(vox, dtype=bool), 1)
    c = v.sum()
    f = sum((v | np.roll(v, 1, a)).sum() for a in range(3))            # a face belongs to <= 2 cubes
    e = 0
    for a in range(3):                                                   # an edge along axis a: 4 cubes around it
        b, d = [(1, 2), (0, 2), (0, 1)][a]
        e += (v | np.roll(v, 1, b) | np.roll(v, 1, d) | np.roll(np.roll(v, 1, b), 1, d)).sum()
    n = v.copy()
    for a in range(3):
        n = n | np.roll(n, 1, a)
    return int(n.sum() - e + f - c)


def test_bars_become_lines_and_rings_stay_rings():
    for shape, sl, length in (((7, 7, 16), (slice(2, 5), slice(2, 5), slice(2, 14)), 10),
                              ((6, 6, 14), (slice(2, 4), slice(2, 4), slice(2, 12)), 10),
                              ((12, 14, 40), (slice(3, 9), slice(4, 10), slice(3, 37)), 30)):
        m = np.zeros(shape, bool)
        m[sl] = True
        s = backend.host_skeletonize_3d(m)
        assert s.sum() == length and ndimage.label(s, S26)[1] == 1
        zz, yy, xx = np.nonzero(s)
        assert len(set(zz)) == 1 and len(set(yy)) == 1                   # a straight line along x
    zz, yy, xx = np.mgrid[:21, :21, :21]
    torus = ((np.sqrt((yy - 10) ** 2 + (xx - 10) ** 2) - 6) ** 2 + (zz - 10) ** 2) <= 6
    ring = backend.host_skeletonize_3d(torus)
    assert ndimage.label(ring, S26)[1] == 1 and euler_characteristic(ring) == 0 == euler_characteristic(torus)
    nb = ndimage.convolve(ring.astype(int), S26.astype(int), mode="constant") - 1
    assert set(nb[ring]) == {2}                                           # a closed curve: two neighbours each


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_topology_is_preserved_and_thinning_is_idempotent(seed):
    rng = np.random.default_rng(seed)
    m = ndimage.gaussian_filter(rng.normal(size=(24, 28, 30)), 2.0) > 0.02
    s = backend.host_skeletonize_3d(m)
    assert s.dtype == bool and not (s & ~m).any() and s.sum() < m.sum() / 4
    assert ndimage.label(s, S26)[1] == ndimage.label(m, S26)[1]
    assert ndimage.label(~np.pad(s, 1))[1] == ndimage.label(~np.pad(m, 1))[1]      # cavities
    assert euler_characteristic(s) == euler_characteristic(m)                      # hence tunnels
    assert np.array_equal(backend.host_skeletonize_3d(s), s)
    # thin: no 2 x 2 x 2 block survives
    blk = ndimage.minimum_filter(s.astype(np.uint8), size=2, mode="constant")
    assert not blk.any()


def test_single_slice_and_2d_input():
    m = np.zeros((1, 30, 30), bool)
    m[0, 5:25, 8:13] = True
    s = backend.host_skeletonize_3d(m)
    assert ndimage.label(s, S26)[1] == 1 and 10 <= s.sum() <= 20
    assert np.array_equal(backend.host_skeletonize_3d(m[0]), s[0])


def test_skeletonize_foreground_is_opt_in_without_scikit_image(monkeypatch):
    """Without scikit-image the option raises like the reference's import would; the library's
    own thinning (not pinned to scikit-image's) serves it only when asked for, and says so."""
    from patchperpix_amd.vote_instances import vote_instances as vi
    m = np.zeros((9, 9, 20), bool)
    m[3:6, 3:6, 2:18] = True
    try:
        import skimage  # noqa: F401
        have = True
    except ImportError:
        have = False
    monkeypatch.delenv("PPP_SKELETONIZE", raising=False)
    if not have:
        with pytest.raises(ImportError, match="skeletonize_backend"):
            vi._skeletonize(m)
    s = vi._skeletonize(m, "ppp")
    assert vi.SKELETONIZE_SERVED_BY == "ppp_host_skeletonize_3d"
    assert s.dtype == bool and s.sum() == 14 and not (s & ~m).any()
    monkeypatch.setenv("PPP_SKELETONIZE", "ppp")
    assert np.array_equal(vi._skeletonize(m), s)
    with pytest.raises(ValueError):
        vi._skeletonize(m, "itk")
