"""Pins the EVALUATOR oracle (oracle/bal_harness.cc: AngleAxisRotatePoint + SnavelyReprojectionError, which the
device evaluator of SURVEY §8 f4 is checked against) to expectations the REFERENCE holds for the functions it restates:

  * internal/ceres/rotation_test.cc:1809-1851  AngleAxis.RotatePointGivesSameAnswerAsRotationMatrix — for theta swept
    over (-pi, pi) and random axes / points, AngleAxisRotatePoint(aa, p) == AngleAxisToRotationMatrix(aa) p to
    kTolerance = 10 eps; the rotation matrix is restated here in numpy from include/ceres/rotation.h:520-566.
  * :1868-1905  AngleAxis.NearZeroRotatePointGivesSameAnswerAsRotationMatrix — the same for |aa| ~ 1e-16 and exactly 0
    (the first-order branch of both functions).
  * examples/snavely_reprojection_error.h:53-105 — the projection formula itself, re-derived independently in numpy
    (rotation by the MATRIX, not by the oracle's Rodrigues code), and its dual-number Jacobian against central differences.
"""
import numpy as np

K_TOLERANCE = np.finfo(np.float64).eps * 10  # rotation_test.cc:61


def angle_axis_to_rotation_matrix(aa):
    """include/ceres/rotation.h:520-566, vectorised; returns (n, 3, 3) with R[i, j] = R(i, j)."""
    aa = np.asarray(aa, dtype=np.float64).reshape(-1, 3)
    # both reference functions take theta = hypot(a0, a1, a2); the oracle (and the device kernel) take sqrt(a0^2 + a1^2 + a2^2) in
    # both places.  What is pinned here is Rodrigues' formula against the matrix formula for ONE theta, so use the oracle's.
    theta = np.sqrt(aa[:, 0] * aa[:, 0] + aa[:, 1] * aa[:, 1] + aa[:, 2] * aa[:, 2])
    R = np.empty((aa.shape[0], 3, 3))
    nz = theta != 0.0
    t = np.where(nz, theta, 1.0)
    wx, wy, wz = aa[:, 0] / t, aa[:, 1] / t, aa[:, 2] / t
    c, s = np.cos(theta), np.sin(theta)
    R[:, 0, 0] = c + wx * wx * (1 - c); R[:, 1, 0] = wz * s + wx * wy * (1 - c); R[:, 2, 0] = -wy * s + wx * wz * (1 - c)
    R[:, 0, 1] = wx * wy * (1 - c) - wz * s; R[:, 1, 1] = c + wy * wy * (1 - c); R[:, 2, 1] = wx * s + wy * wz * (1 - c)
    R[:, 0, 2] = wy * s + wx * wz * (1 - c); R[:, 1, 2] = -wx * s + wy * wz * (1 - c); R[:, 2, 2] = c + wz * wz * (1 - c)
    z = ~nz  # first-order Taylor expansion at zero (:555-565)
    if z.any():
        a = aa[z]
        Rz = np.zeros((a.shape[0], 3, 3))
        Rz[:, 0, 0] = Rz[:, 1, 1] = Rz[:, 2, 2] = 1.0
        Rz[:, 1, 0] = a[:, 2]; Rz[:, 2, 0] = -a[:, 1]; Rz[:, 0, 1] = -a[:, 2]; Rz[:, 2, 1] = a[:, 0]; Rz[:, 0, 2] = a[:, 1]; Rz[:, 1, 2] = -a[:, 0]
        R[z] = Rz
    return R


def test_rotate_point_gives_same_answer_as_rotation_matrix(oracle):
    rng = np.random.default_rng(0)
    i = np.arange(10000)
    theta = np.repeat((2.0 * i * 0.0011 - 1.0) * np.pi, 50)           # the reference's sweep, :1819-1821
    aa = rng.uniform(-1.0, 1.0, (theta.shape[0], 3))
    p = rng.uniform(-1.0, 1.0, (theta.shape[0], 3))
    aa *= (theta / np.linalg.norm(aa, axis=1))[:, None]
    want = np.einsum("nij,nj->ni", angle_axis_to_rotation_matrix(aa), p)
    got = oracle.angle_axis_rotate_points(aa, p)
    # the reference's two functions share one theta = hypot(...) and agree to 10 eps; here the C side contracts a0^2 + a1^2 + a2^2
    # into FMAs and numpy does not, so the two thetas may differ by an ulp — which moves a rotated point by |p| ulp(theta),
    # up to 1.4e-14 at the sweep's end (|theta| = 21 pi).  Allow exactly that on top of the reference's tolerance.
    tol = K_TOLERANCE + 2.0 * np.spacing(np.abs(theta))[:, None] * np.linalg.norm(p, axis=1)[:, None]
    assert (np.abs(got - want) <= tol).all(), np.abs(got - want).max()
    small = np.abs(theta) < 1.0   # where ulp(theta) is below eps the reference's own tolerance holds as is
    assert np.abs(got[small] - want[small]).max() <= K_TOLERANCE


def test_near_zero_rotate_point_gives_same_answer_as_rotation_matrix(oracle):
    rng = np.random.default_rng(1)
    n = 10000
    aa = rng.uniform(-1.0, 1.0, (n, 3))
    p = rng.uniform(-1.0, 1.0, (n, 3))
    theta = (2.0 * np.arange(n) * 0.0001 - 1.0) * 1e-16               # :1878, includes theta == 0 exactly at i = 5000
    aa *= (theta / np.linalg.norm(aa, axis=1))[:, None]
    assert (aa[5000] == 0).all()
    want = np.einsum("nij,nj->ni", angle_axis_to_rotation_matrix(aa), p)
    got = oracle.angle_axis_rotate_points(aa, p)
    assert np.abs(got - want).max() <= K_TOLERANCE
    # exactly zero rotation is the identity
    np.testing.assert_array_equal(oracle.angle_axis_rotate_points(np.zeros((4, 3)), p[:4]), p[:4])


def snavely_reference(cam, pt, obs):
    """examples/snavely_reprojection_error.h:53-105 with the rotation done by the MATRIX."""
    R = angle_axis_to_rotation_matrix(cam[:, :3])
    q = np.einsum("nij,nj->ni", R, pt) + cam[:, 3:6]
    xp, yp = -q[:, 0] / q[:, 2], -q[:, 1] / q[:, 2]
    r2 = xp * xp + yp * yp
    dist = 1.0 + r2 * (cam[:, 7] + cam[:, 8] * r2)
    return np.stack([cam[:, 6] * dist * xp - obs[:, 0], cam[:, 6] * dist * yp - obs[:, 1]], 1)


def test_snavely_residual_and_jacobian(oracle):
    rng = np.random.default_rng(2)
    n = 2000
    cam = np.concatenate([rng.uniform(-1, 1, (n, 3)) * rng.uniform(0, 2.5, (n, 1)), rng.uniform(-1, 1, (n, 3)), 500 + 500 * rng.random((n, 1)),
                          1e-2 * rng.standard_normal((n, 1)), 1e-3 * rng.standard_normal((n, 1))], 1)
    cam[:5, :3] = 0.0                                                 # the zero-rotation branch
    pt = rng.uniform(-1, 1, (n, 3)) + np.array([0.0, 0.0, -6.0])       # in front of the camera (BAL looks down -z)
    obs = rng.uniform(-50, 50, (n, 2))
    # keep the triples whose point stays well off the camera's z = 0 plane (the projection divides by z)
    qz = (np.einsum("nij,nj->ni", angle_axis_to_rotation_matrix(cam[:, :3]), pt) + cam[:, 3:6])[:, 2]
    keep = np.abs(qz) > 2.0
    keep[:5] = True
    cam, pt, obs = cam[keep], pt[keep], obs[keep]
    assert cam.shape[0] > 1000
    r, jc, jp = oracle.snavely_batch(cam, pt, obs)
    want = snavely_reference(cam, pt, obs)
    assert np.abs(r - want).max() <= 1e-11 * np.abs(want).max()
    # Jacobian by central differences of the independent formula (cameras with a non-zero rotation: the zero branch is not
    # differentiable through the matrix formula's own switch)
    sel = slice(5, None)
    h = 1e-6
    for k in range(9):
        d = np.zeros(9); d[k] = h
        fd = (snavely_reference(cam[sel] + d, pt[sel], obs[sel]) - snavely_reference(cam[sel] - d, pt[sel], obs[sel])) / (2 * h)
        assert np.abs(jc[sel, :, k] - fd).max() <= 2e-6 * max(np.abs(fd).max(), 1.0), k
    for k in range(3):
        d = np.zeros(3); d[k] = h
        fd = (snavely_reference(cam[sel], pt[sel] + d, obs[sel]) - snavely_reference(cam[sel], pt[sel] - d, obs[sel])) / (2 * h)
        assert np.abs(jp[sel, :, k] - fd).max() <= 2e-6 * max(np.abs(fd).max(), 1.0), k
