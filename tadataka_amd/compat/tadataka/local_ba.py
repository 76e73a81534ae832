"""tadataka.local_ba (reference tadataka/local_ba.py:14-178).

Projection.compute / .jacobians evaluate every observation in ONE device launch
(the reference loops in Python with three Cython calls per observation), and
`block_sums` returns the fused residual + Jacobian + per-pose / per-point block
reduction that the sparse bundle-adjustment solve starts from.

The Levenberg-Marquardt driver LocalBundleAdjustment wraps the third-party
`sparseba` Schur solver in the reference (local_ba.py:72,77); here the Schur
complement step runs on the device as well (tdk_ba_step, standard additive
damping -- sparseba's exact damping convention is not pinned)."""
import numpy as np

from tadataka_amd import ops


class Projection(object):
    def __init__(self, viewpoint_indices, point_indices):
        assert(len(viewpoint_indices) == len(point_indices))
        self.viewpoint_indices = np.ascontiguousarray(viewpoint_indices, dtype=np.int64)
        self.point_indices = np.ascontiguousarray(point_indices, dtype=np.int64)
        self.n_visible = len(self.point_indices)

    def compute(self, poses, points):
        """x_pred [n_visible, 2] = transform_project(poses[j], points[i])."""
        return ops.ba_projection(poses, points, self.viewpoint_indices, self.point_indices,
                                 jacobians=False)

    def jacobians(self, poses, points):
        """A [n_visible, 2, 6] = d x / d pose, B [n_visible, 2, 3] = d x / d point."""
        _, A, B = ops.ba_projection(poses, points, self.viewpoint_indices, self.point_indices)
        return A, B

    def block_sums(self, poses, points, x_true):
        """U_j = sum A^T A, ea_j = sum A^T e, V_i = sum B^T B, eb_i = sum B^T e
        (e = x_true - x_pred) and sum ||e||^2, fused on the device."""
        U, ea, V, eb, err = ops.ba_block_reduce(poses, points, x_true, self.viewpoint_indices,
                                                self.point_indices)
        iu6, iu3 = np.triu_indices(6), np.triu_indices(3)
        Um = np.zeros((U.shape[0], 6, 6)); Um[:, iu6[0], iu6[1]] = U; Um[:, iu6[1], iu6[0]] = U
        Vm = np.zeros((V.shape[0], 3, 3)); Vm[:, iu3[0], iu3[1]] = V; Vm[:, iu3[1], iu3[0]] = V
        return Um, ea, Vm, eb, err


def calc_relative_error(current_error, new_error):
    return np.abs((current_error - new_error) / new_error)


def calc_errors(x_true, x_pred):
    return np.sum(np.power(x_true - x_pred, 2), axis=1)


def calc_error(x_true, x_pred):
    return np.mean(calc_errors(x_true, x_pred))


class LocalBundleAdjustment(object):
    """Levenberg-Marquardt bundle adjustment (reference local_ba.py:60-134).

    The reference obtains each update from the third-party `sparseba.SBA`
    (per-observation Jacobians in, Schur solve on the CPU); here one device call
    (tdk_ba_step) computes residuals, Jacobians, the block sums, the Schur
    complement and the back-substitution for a given damping mu."""

    def __init__(self, viewpoint_indices, point_indices, x_true):
        assert(len(viewpoint_indices) == x_true.shape[0])
        assert(len(point_indices) == x_true.shape[0])
        self.projection = Projection(viewpoint_indices, point_indices)
        self.x_true = x_true
        self._graph = None

    def _device_graph(self, poses, points):
        if self._graph is None:
            self._graph = ops.BundleAdjustment(len(poses), len(points),
                                               self.projection.viewpoint_indices,
                                               self.projection.point_indices, self.x_true)
        return self._graph

    def calc_update(self, poses, points, mu):
        dposes, dpoints, _ = self._device_graph(poses, points).step(poses, points, mu)
        return dposes, dpoints

    def calc_error(self, poses, points):
        g = self._device_graph(poses, points)
        return g.sum_squared_error(poses, points) / g.n

    def calc_new_error(self, poses, points, mu):
        dposes, dpoints = self.calc_update(poses, points, mu)
        return dposes, dpoints, self.calc_error(poses + dposes, points + dpoints)

    def lm_update(self, poses, points, mu, nu):
        error0 = self.calc_error(poses, points)
        for trial_mu in (mu / nu, mu):
            dposes, dpoints, error = self.calc_new_error(poses, points, trial_mu)
            if error < error0:
                return poses + dposes, points + dpoints, trial_mu, error
        error, new_mu = np.inf, mu
        while error > error0:
            new_mu = new_mu * nu
            dposes, dpoints, error = self.calc_new_error(poses, points, new_mu)
        return poses + dposes, points + dpoints, new_mu, error

    def compute(self, initial_rotvecs, initial_translations, initial_points,
                max_iter=200, initial_mu=1.0, nu=100.0,
                absolute_error_threshold=1e-8, relative_error_threshold=1e-6):
        poses = np.hstack((initial_rotvecs, initial_translations))
        points = np.asarray(initial_points, dtype=np.float64)
        # the loop of the reference (lm_update per iteration, :115-134) runs inside
        # the library with the parameters resident on the device (tdk_ba_solve);
        # its per-iteration report is printed from the returned error history
        poses, points, errors = self._device_graph(poses, points).solve(
            poses, points, max_iter=max_iter, initial_mu=initial_mu, nu=nu,
            absolute_error_threshold=absolute_error_threshold,
            relative_error_threshold=relative_error_threshold)
        for iter_ in range(len(errors) - 1):
            print(f"absolute_error[{iter_}] = {errors[iter_ + 1]}")
            print(f"relative_error[{iter_}] = {calc_relative_error(errors[iter_], errors[iter_ + 1])}")
        return poses[:, 0:3], poses[:, 3:6], points


def run_ba(viewpoint_indices, point_indices, poses, points, keypoints_true):
    """Poses in / out are tadataka.pose.Pose objects (reference local_ba.py:137-152)."""
    from scipy.spatial.transform import Rotation
    from tadataka.pose import Pose
    ba = LocalBundleAdjustment(viewpoint_indices, point_indices, keypoints_true)
    rotvecs = np.array([p.rotation.as_rotvec() for p in poses])
    ts = np.array([p.t for p in poses])
    rotvecs, ts, points = ba.compute(rotvecs, ts, points, absolute_error_threshold=1e-9,
                                     max_iter=5, relative_error_threshold=0.20)
    return [Pose(Rotation.from_rotvec(r), t) for r, t in zip(rotvecs, ts)], points


def can_run_ba(n_viewpoints, n_points, n_visible, n_pose_params, n_point_params):
    """sparseba.can_run_ba (the package is absent here; its published rule): J^T J cannot be invertible with fewer
    rows (two per observation) than columns (pose and point parameters)."""
    return 2 * n_visible >= n_pose_params * n_viewpoints + n_point_params * n_points


def test_unique(viewpoint_indices, point_indices):
    """reference local_ba.py:155-157: no (viewpoint, point) pair observed twice."""
    A = np.vstack((viewpoint_indices, point_indices))
    assert(np.unique(A, axis=1).shape[1] == A.shape[1])


test_unique.__test__ = False      # a helper of the reference's API, not a test of this repository


def try_run_ba(viewpoint_indices, point_indices, poses, points, keypoints_true):
    """reference local_ba.py:160-179 (called by tadataka/vo/feature_based.py:226): run_ba if the graph can
    determine its parameters, else a RuntimeWarning and the inputs back."""
    import warnings
    assert(len(viewpoint_indices) == len(point_indices))
    assert(len(set(viewpoint_indices)) == len(poses))
    assert(len(set(point_indices)) == len(points))
    test_unique(viewpoint_indices, point_indices)
    if not can_run_ba(n_viewpoints=len(poses), n_points=len(points), n_visible=len(keypoints_true),
                      n_pose_params=6, n_point_params=3):
        warnings.warn("Arguments are not satisfying condition to run BA", RuntimeWarning)
        return poses, points
    return run_ba(viewpoint_indices, point_indices, poses, points, keypoints_true)
