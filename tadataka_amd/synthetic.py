"""Deterministic synthetic frames for parity tests and bench.py.

NumPy only.  A frame pair is rendered analytically from a smooth texture so
that the true SE(3) between the two frames is known:

    texture(x, y) = 0.5 + 0.25 sin(x/7) cos(y/5) + 0.2 sin((x+y)/11)
    D0(x, y)      = 2.0 + 0.3 sin(x/40) + 0.2 cos(y/30)
    I1 = texture(pixel grid) + noise,  I0(u0) = texture(warp(u0)) + noise
"""
import numpy as np


def texture(x, y, period_scale=1.0):
    s = period_scale
    return (0.5 + 0.25 * np.sin(x / (7.0 * s)) * np.cos(y / (5.0 * s))
            + 0.2 * np.sin((x + y) / (11.0 * s)))


def depth_map(x, y):
    return 2.0 + 0.3 * np.sin(x / 40.0) + 0.2 * np.cos(y / 30.0)


def camera_for(width, height):
    """cam = (fx, fy, ox, oy): f = 525 * W/640, o = (W/2, H/2)."""
    f = 525.0 * width / 640.0
    return np.array([f, f, width / 2.0, height / 2.0])


def rodrigues(rotvec):
    rotvec = np.asarray(rotvec, dtype=np.float64)
    theta = np.linalg.norm(rotvec)
    K = np.array([[0., -rotvec[2], rotvec[1]],
                  [rotvec[2], 0., -rotvec[0]],
                  [-rotvec[1], rotvec[0], 0.]])
    if theta < 1e-12:
        return np.eye(3) + K + 0.5 * K @ K
    return (np.eye(3) + np.sin(theta) / theta * K
            + (1 - np.cos(theta)) / theta ** 2 * (K @ K))


def random_pose(rng, rot_scale=0.005, trans_scale=0.01):
    omega = rng.uniform(-rot_scale, rot_scale, 3)
    t = rng.uniform(-trans_scale, trans_scale, 3)
    return omega, t


def make_pair(height=480, width=640, seed=0, noise=0.02, rot_scale=0.005,
              trans_scale=0.01, period_scale=1.0):
    """Returns dict(I0, D0, I1, cam, omega, t, T10)."""
    rng = np.random.default_rng(seed)
    omega, t = random_pose(rng, rot_scale, trans_scale)
    R = rodrigues(omega)
    cam = camera_for(width, height)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    D0 = depth_map(xs, ys)
    I1 = texture(xs, ys, period_scale) + noise * rng.uniform(-1, 1, (height, width))
    # warp the grid of frame 0 into frame 1 and sample the analytic texture
    xn = (xs - cam[2]) / cam[0]
    yn = (ys - cam[3]) / cam[1]
    P0 = np.stack([xn * D0, yn * D0, D0], axis=-1)
    P1 = P0 @ R.T + t
    u1x = P1[..., 0] / P1[..., 2] * cam[0] + cam[2]
    u1y = P1[..., 1] / P1[..., 2] * cam[1] + cam[3]
    I0 = texture(u1x, u1y, period_scale) + noise * rng.uniform(-1, 1, (height, width))
    T10 = np.eye(4)
    T10[:3, :3] = R
    T10[:3, 3] = t
    return dict(I0=np.ascontiguousarray(I0), D0=np.ascontiguousarray(D0),
                I1=np.ascontiguousarray(I1), cam=cam, omega=omega, t=t, T10=T10)


def make_semi_dense_case(height=480, width=640, seed=1, valid_fraction=0.3,
                         baseline=(0.1, 0.0, 0.0), noise=0.0):
    """Key/ref frames with a pure-translation baseline, a Bernoulli age map,
    a prior depth = GT * U(0.9, 1.1) and prior variance 0.05."""
    rng = np.random.default_rng(seed)
    cam = camera_for(width, height)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    depth_key = depth_map(xs, ys)
    T_wk = np.eye(4)
    T_wr = np.eye(4)
    T_wr[:3, 3] = np.asarray(baseline, dtype=np.float64)
    key_image = texture(xs, ys) + noise * rng.uniform(-1, 1, (height, width))
    # ref image: texture is attached to the key-frame surface; sample it where
    # the ref pixel's ray meets (approximately) that surface: use key depth as
    # a smooth proxy so that the images are consistent to first order.
    xn = (xs - cam[2]) / cam[0]
    yn = (ys - cam[3]) / cam[1]
    Pr = np.stack([xn * depth_key, yn * depth_key, depth_key], axis=-1)
    Pk = Pr + T_wr[:3, 3]          # ref -> world(=key) for identity rotations
    ukx = Pk[..., 0] / Pk[..., 2] * cam[0] + cam[2]
    uky = Pk[..., 1] / Pk[..., 2] * cam[1] + cam[3]
    ref_image = texture(ukx, uky) + noise * rng.uniform(-1, 1, (height, width))
    age = (rng.uniform(0, 1, (height, width)) < valid_fraction).astype(np.uint64)
    prior_depth = depth_key * rng.uniform(0.9, 1.1, (height, width))
    prior_variance = np.full((height, width), 0.05)
    return dict(cam=cam, key_image=np.ascontiguousarray(key_image),
                ref_image=np.ascontiguousarray(ref_image), T_wk=T_wk, T_wr=T_wr,
                age=age, prior_depth=np.ascontiguousarray(prior_depth),
                prior_variance=prior_variance, depth_gt=depth_key)


def make_track(height, width, n_frames, step=(0.03, 0.005, 0.01)):
    """A camera sliding by `step` per frame in front of the textured surface of frame 0
    (first-order consistent views: the texture is attached to that surface).  Returns
    (cam, depth of frame 0, [T_wf], [image]) -- the frames of a short semi-dense track."""
    cam = camera_for(width, height)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    depth0 = depth_map(xs, ys)
    xn, yn = (xs - cam[2]) / cam[0], (ys - cam[3]) / cam[1]
    transforms, images = [], []
    for k in range(n_frames):
        T = np.eye(4)
        T[:3, 3] = [step[0] * k, step[1] * k, step[2] * k]
        P = np.stack([xn * depth0, yn * depth0, depth0], axis=-1) + T[:3, 3]
        images.append(np.ascontiguousarray(texture(P[..., 0] / P[..., 2] * cam[0] + cam[2],
                                                   P[..., 1] / P[..., 2] * cam[1] + cam[3])))
        transforms.append(T)
    return cam, depth0, transforms, images


def make_ba_case(n_poses=8, n_points=50000, seed=5, perturb=1e-3):
    """all points visible from all poses."""
    rng = np.random.default_rng(seed)
    omegas = 0.1 * rng.uniform(-1, 1, (n_poses, 3))
    ts = rng.uniform(-1, 1, (n_poses, 3))
    poses = np.hstack([omegas, ts])
    points = np.column_stack([rng.uniform(-5, 5, n_points), rng.uniform(-5, 5, n_points),
                              rng.uniform(4, 12, n_points)])
    vp_idx = np.repeat(np.arange(n_poses, dtype=np.int64), n_points)
    pt_idx = np.tile(np.arange(n_points, dtype=np.int64), n_poses)
    poses_noisy = poses + perturb * rng.uniform(-1, 1, poses.shape)
    points_noisy = points + perturb * rng.uniform(-1, 1, points.shape)
    return dict(poses=poses, points=points, poses_noisy=poses_noisy,
                points_noisy=points_noisy, vp_idx=vp_idx, pt_idx=pt_idx)
