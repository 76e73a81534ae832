#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ by running the
REFERENCE itself in the build container (it needs /root/reference and is never
run on the GPU box -- only the .npz outputs travel).

What runs unmodified from the reference (imported from /root/reference):
  tadataka.vo.dvo (_PoseChangeEstimator, PoseChangeEstimator, calc_pose_update),
  tadataka.metric, tadataka.warp, tadataka.pose, tadataka.se3, tadataka.so3,
  tadataka.math, tadataka.robust.weights, tadataka.vo.dvo.jacobian,
  tadataka.coordinates, tadataka.utils, tadataka.projection,
  tadataka.rigid_transform, tadataka.matrix, tadataka.interpolation,
  tadataka.camera, tadataka.irls, and the Cython tadataka.transform_project
  built from the reference's own sympy generator (so3_codegen.generate()).

What is stubbed (absent toolchains/packages): the Rust leaves
`rust_bindings.*` and `tadataka.camera._normalizer` are replaced by a
vectorised NumPy restatement written here (independent of oracle/tdk_oracle.c),
except `interpolation`, which calls the reference's own _bilinear.cpp compiled
as oracle/_ref/libref_bilinear.so; scikit-image's `rescale` is replaced by the
build's bilinear pyramid (parity unpinned for that step).

Usage:  python tests/golden/generate_golden.py
"""
import ctypes
import importlib
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from tadataka_amd import synthetic  # noqa: E402
from oracle import oracle as orc    # noqa: E402  (only for the pyramid stand-in)


# ---------------------------------------------------------------------------
# stubs
# ---------------------------------------------------------------------------
def _load_ref_bilinear():
    so = os.path.join(REPO, "oracle", "_ref", "libref_bilinear.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "ref"])
    lib = ctypes.CDLL(so)
    fn = lib._Z14_interpolationPKdiS0_iPd
    dp = ctypes.POINTER(ctypes.c_double)

    def interpolation(image, coords):
        image = np.ascontiguousarray(image, dtype=np.float64)
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        out = np.empty(coords.shape[0])
        fn(image.ctypes.data_as(dp), ctypes.c_int(image.shape[1]),
           coords.ctypes.data_as(dp), ctypes.c_int(coords.shape[0]),
           out.ctypes.data_as(dp))
        return out
    return interpolation


ANTI_ALIASING = [False]


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def rescale(image, scale, **kw):
        # ANTI_ALIASING[0]: the oracle's restatement of skimage's default Gaussian prefilter
        return orc.rescale(image, scale, anti_aliasing=ANTI_ALIASING[0])

    mod("skimage")
    mod("skimage.transform", rescale=rescale, resize=None,
        ProjectiveTransform=object, FundamentalMatrixTransform=object)
    mod("skimage.color", rgb2gray=None)
    mod("cv2")
    mod("tadataka.feature", empty_match=None)

    # src/camera.rs:36-48, tadataka/camera/_normalizer.cpp:16-17,25-26
    mod("tadataka.camera._normalizer",
        normalize=lambda kp, f, o: (np.asarray(kp, dtype=np.float64) - o) / f,
        unnormalize=lambda kp, f, o: np.asarray(kp, dtype=np.float64) * f + o)

    def project_vecs(P):            # src/projection.rs:11-14
        return P[:, 0:2] / (P[:, [2]] + 1e-16)

    def inv_project_vecs(xs, d):    # src/projection.rs:16-18
        return np.column_stack((xs[:, 0] * d, xs[:, 1] * d, d))

    def transform(T, P):            # src/transform.rs:17-23
        Ph = np.column_stack((P, np.ones(P.shape[0])))
        return (T @ Ph.T).T[:, 0:3]

    def warp_vecs(T10, xs, depths):  # src/warp.rs:39-49
        P1 = transform(T10, inv_project_vecs(xs, depths))
        return project_vecs(P1), P1[:, 2].copy()

    rb = mod("rust_bindings")
    rb.warp = mod("rust_bindings.warp", warp_vecs=warp_vecs,
                  warp_vec=lambda T, x, d: tuple(a[0] for a in warp_vecs(T, x[None], np.array([d]))))
    rb.interpolation = mod("rust_bindings.interpolation", interpolation=_load_ref_bilinear())
    rb.projection = mod("rust_bindings.projection", project_vecs=project_vecs,
                        project_vec=lambda p: project_vecs(p[None])[0],
                        inv_project_vecs=inv_project_vecs,
                        inv_project_vec=lambda x, d: inv_project_vecs(x[None], np.array([d]))[0])
    rb.transform = mod("rust_bindings.transform", transform=transform)
    rb.homogeneous = mod(
        "rust_bindings.homogeneous",
        to_homogeneous_vecs=lambda X: np.column_stack((X, np.ones(X.shape[0]))),
        to_homogeneous_vec=lambda x: np.append(x, 1.0))
    rb.triangulation = mod("rust_bindings.triangulation", calc_depth0=None)

    sys.path.insert(0, REF)
    vo = types.ModuleType("tadataka.vo")
    vo.__path__ = [os.path.join(REF, "tadataka", "vo")]
    import tadataka  # noqa: F401  (reference package)
    sys.modules["tadataka.vo"] = vo


def build_transform_project():
    """The reference's own recipe (setup.py:14-15,26-35): sympy -> C, cythonize."""
    scratch = tempfile.mkdtemp(prefix="tdk_golden_")
    os.makedirs(os.path.join(scratch, "tadataka", "_transform_project"))
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        from tadataka import so3_codegen
        so3_codegen.generate()
        shutil.copy(os.path.join(REF, "tadataka", "transform_project.pyx"),
                    os.path.join(scratch, "tadataka", "transform_project.pyx"))
        setup_py = os.path.join(scratch, "setup_tp.py")
        with open(setup_py, "w") as f:
            f.write(
                "from setuptools import setup, Extension\n"
                "from Cython.Build import cythonize\nimport numpy as np\n"
                "ext = Extension('refbuild.transform_project',\n"
                "  sources=['tadataka/transform_project.pyx',\n"
                "           'tadataka/_transform_project/_transform_project.c',\n"
                "           'tadataka/_transform_project/_pose_jacobian.c',\n"
                "           'tadataka/_transform_project/_point_jacobian.c',\n"
                "           'tadataka/_transform_project/_exp_so3.c'],\n"
                "  include_dirs=[np.get_include()], extra_compile_args=['-O2'])\n"
                "ext.cython_directives = {'language_level': 3}\n"
                "setup(name='refbuild', ext_modules=cythonize([ext]))\n")
        os.makedirs(os.path.join(scratch, "refbuild"), exist_ok=True)
        open(os.path.join(scratch, "refbuild", "__init__.py"), "w").close()
        subprocess.check_call([sys.executable, setup_py, "build_ext", "--inplace"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sys.path.insert(0, scratch)
        return importlib.import_module("refbuild.transform_project")
    finally:
        os.chdir(cwd)


# ---------------------------------------------------------------------------
# fixtures
# ---------------------------------------------------------------------------
def capture_dvo(h, w, seed, weights_list, tag, keep_rows):
    """Runs the reference _PoseChangeEstimator on a synthetic pair and records
    every calc_pose_update / PhotometricError call."""
    import tadataka.vo.dvo as dvo
    from tadataka.camera import CameraModel, CameraParameters
    from tadataka.pose import Pose

    pair = synthetic.make_pair(h, w, seed=seed)
    cam = pair["cam"]
    cm = CameraModel(CameraParameters(cam[0:2], cam[2:4]), distortion_model=None)
    out = dict(I0=pair["I0"], D0=pair["D0"], I1=pair["I1"], cam=cam,
               omega_true=pair["omega"], t_true=pair["t"])

    rng = np.random.default_rng(100 + seed)
    wmap = rng.uniform(0.2, 2.0, (h, w))
    out["weight_map"] = wmap

    for wname in weights_list:
        weights = wmap if wname == "map" else wname
        rec = dict(updates=[], errors=[])

        orig_solve = dvo.solve_linear_equation
        orig_err_cls = dvo.PhotometricError

        def solve(J, r, weights=None, **kw):
            xi = orig_solve(J, r, weights, **kw)
            rec["updates"].append((J.copy(), r.copy(),
                                   None if weights is None else weights.copy(), xi.copy()))
            return xi

        class Err(orig_err_cls):
            def __call__(self, pose10):
                e = super().__call__(pose10)
                rec["errors"].append((pose10.T.copy(), e))
                return e

        dvo.solve_linear_equation = solve
        dvo.PhotometricError = Err
        try:
            est = dvo._PoseChangeEstimator(cm, cm, max_iter=20)
            pose = est(pair["I0"], pair["D0"], pair["I1"], Pose.identity(), weights)
        finally:
            dvo.solve_linear_equation = orig_solve
            dvo.PhotometricError = orig_err_cls

        key = f"{tag}_{wname}"
        out[f"{key}_final_rotvec"] = pose.rotation.as_rotvec()
        out[f"{key}_final_t"] = pose.t
        out[f"{key}_n_updates"] = len(rec["updates"])
        out[f"{key}_err_T"] = np.array([e[0] for e in rec["errors"]])
        out[f"{key}_err_val"] = np.array([e[1] for e in rec["errors"]])
        for k, (J, r, wv, xi) in enumerate(rec["updates"]):
            sw = np.ones(len(r)) if wv is None else wv
            out[f"{key}_u{k}_n_valid"] = J.shape[0]
            out[f"{key}_u{k}_xi"] = xi
            # weighted normal equations in float64 (what the GPU reduces to)
            out[f"{key}_u{k}_H"] = (J * sw[:, None]).T @ J
            out[f"{key}_u{k}_b"] = (J * sw[:, None]).T @ r
            if keep_rows:
                # J and r do not depend on the weight mode for the same input
                # pose; keep the rows once (weights=None) and w for every mode
                if wname is None:
                    out[f"{key}_u{k}_J"] = J
                    out[f"{key}_u{k}_r"] = r
                out[f"{key}_u{k}_w"] = sw
            # the pose this update was evaluated at = pose of the matching
            # PhotometricError call (errors[k] is evaluated before update k)
    return out


def capture_pyramid(h, w, seed):
    import tadataka.vo.dvo as dvo
    from tadataka.camera import CameraModel, CameraParameters
    pair = synthetic.make_pair(h, w, seed=seed)
    cam = pair["cam"]
    cm = CameraModel(CameraParameters(cam[0:2], cam[2:4]), distortion_model=None)
    out = dict(I0=pair["I0"], D0=pair["D0"], I1=pair["I1"], cam=cam,
               omega_true=pair["omega"], t_true=pair["t"])
    # plain bilinear pyramid ("pyr_") and the anti-aliased one skimage builds by default ("pyr_aa_")
    for tag, aa in (("pyr", False), ("pyr_aa", True)):
        ANTI_ALIASING[0] = aa
        for wname in (None, "huber"):
            est = dvo.PoseChangeEstimator(cm, cm, n_coarse_to_fine=3, max_iter=20)
            pose = est(pair["I0"], pair["D0"], pair["I1"], wname)
            out[f"{tag}_{wname}_rotvec"] = pose.rotation.as_rotvec()
            out[f"{tag}_{wname}_t"] = pose.t
    ANTI_ALIASING[0] = False
    return out


def capture_vga_pyramid(seed=0):
    """BASELINE configs[1] end to end: the reference's own PoseChangeEstimator
    (tadataka/vo/dvo/__init__.py:114-150) at 640x480, n_coarse_to_fine=3,
    layer_size_ratio=1.5, max_iter=20 on the seed-0 synthetic pair -- final pose,
    the pose after every level and the number of PhotometricError evaluations per
    level (= updates + 1).  Inputs are regenerated from the seed by the tests."""
    import tadataka.vo.dvo as dvo
    from tadataka.camera import CameraModel, CameraParameters
    pair = synthetic.make_pair(480, 640, seed=seed)
    cam = pair["cam"]
    cm = CameraModel(CameraParameters(cam[0:2], cam[2:4]), distortion_model=None)
    out = dict(cam=cam, omega_true=pair["omega"], t_true=pair["t"])
    orig_err_cls = dvo.PhotometricError
    orig_level = dvo._PoseChangeEstimator.__call__
    for tag, aa in (("pyr", False), ("pyr_aa", True)):
        ANTI_ALIASING[0] = aa
        for wname in (None, "huber"):
            counts, level_poses = [], []

            class Err(orig_err_cls):
                def __init__(self, *a, **k):
                    super().__init__(*a, **k)
                    counts.append(0)

                def __call__(self, pose10):
                    counts[-1] += 1
                    return super().__call__(pose10)

            def level_call(self, I0, D0, I1, pose10, weights=None):
                pose = orig_level(self, I0, D0, I1, pose10, weights)
                level_poses.append(np.concatenate([pose.rotation.as_rotvec(), pose.t]))
                return pose

            dvo.PhotometricError = Err
            dvo._PoseChangeEstimator.__call__ = level_call
            try:
                est = dvo.PoseChangeEstimator(cm, cm, n_coarse_to_fine=3, max_iter=20)
                pose = est(pair["I0"], pair["D0"], pair["I1"], wname)
            finally:
                dvo.PhotometricError = orig_err_cls
                dvo._PoseChangeEstimator.__call__ = orig_level
            out[f"{tag}_{wname}_rotvec"] = pose.rotation.as_rotvec()
            out[f"{tag}_{wname}_t"] = pose.t
            out[f"{tag}_{wname}_evals"] = np.array(counts, dtype=np.int64)        # coarse -> fine
            out[f"{tag}_{wname}_level_poses"] = np.array(level_poses)             # coarse -> fine
    ANTI_ALIASING[0] = False
    return out


def capture_pyref():
    """Outputs of the reference's pure-Python numerics on seeded inputs."""
    from tadataka.robust import weights as rw
    from tadataka import se3, math as tmath, irls
    from tadataka.vo.dvo.jacobian import calc_jacobian, calc_image_gradient
    from tadataka.utils import is_in_image_range
    from tadataka.coordinates import image_coordinates
    from tadataka.pose import Pose
    rng = np.random.default_rng(7)
    out = {}
    r = rng.normal(0, 0.7, 2001)
    r[::50] *= 8
    out["w_r"] = r
    out["w_huber"] = rw.compute_weights_huber(r)
    out["w_student_t"] = rw.compute_weights_student_t(r)
    out["w_tukey"] = rw.compute_weights_tukey(r)
    xis = np.vstack([rng.uniform(-1, 1, (20, 6)), np.zeros((1, 6)),
                     np.concatenate([rng.uniform(-1, 1, 3), [0, 0, 0]])[None]])
    out["se3_xi"] = xis
    out["se3_t"] = np.array([se3.exp_se3_t_(x) for x in xis])
    out["se3_G"] = np.array([se3.exp_se3(x) for x in xis])
    # Pose.from_se3(xi) * Pose(prior)
    prior = Pose.from_se3(rng.uniform(-0.3, 0.3, 6))
    comp = [Pose.from_se3(x) * prior for x in xis]
    out["pose_prior_T"] = prior.T
    out["pose_comp_T"] = np.array([p.T for p in comp])
    A = rng.normal(size=(300, 6)); b = rng.normal(size=300); w = rng.uniform(0.1, 2, 300)
    out["ls_A"], out["ls_b"], out["ls_w"] = A, b, w
    out["ls_x"] = tmath.solve_linear_equation(A, b)
    out["ls_xw"] = tmath.solve_linear_equation(A, b, w)
    P = rng.uniform(-10, 10, (64, 3)); gx = rng.uniform(-1, 1, 64); gy = rng.uniform(-1, 1, 64)
    out["jac_P"], out["jac_gx"], out["jac_gy"] = P, gx, gy
    out["jac_J"] = calc_jacobian([300., 400.], gx, gy, P)
    img = rng.uniform(0, 1, (9, 13))
    out["grad_img"] = img
    out["grad_gx"], out["grad_gy"] = calc_image_gradient(img)
    kp = np.array([[0., 0.], [12., 8.], [12.01, 3.], [-0.01, 2.], [5.5, 8.0], [3., 8.0001]])
    out["rng_kp"] = kp
    out["rng_mask"] = is_in_image_range(kp, (9, 13))
    out["coords_3x4"] = image_coordinates((3, 4))
    # irls.fit, p = 3 (its only call site, flow_estimation.py:10-12)
    X = np.column_stack([rng.normal(size=(400, 2)), np.ones(400)])
    beta = np.array([0.7, -1.3, 0.2])
    y = X @ beta + 0.05 * rng.normal(size=400)
    y[::17] += 4.0
    out["irls_X"], out["irls_y"] = X, y
    out["irls_beta"] = irls.fit(X, y)
    return out


def capture_ba(tp):
    rng = np.random.default_rng(11)
    n = 1200
    omegas = rng.uniform(-1, 1, (n, 3)) * rng.choice([0.01, 0.3, 1.0, 2.5], (n, 1))
    special = [np.zeros(3), np.array([1e-9, 0, 0]), np.array([0, 1e-9, -1e-9]),
               np.array([np.pi / 2, 0, 0]), np.array([0, -np.pi / 2, 0]),
               np.array([0, 0, np.pi]), np.array([-np.pi, 0, 0]),
               np.pi * np.array([0.6, 0.0, 0.8]), np.array([1e-5, 2e-5, -1e-5])]
    omegas[:len(special)] = special
    ts = rng.uniform(-2, 2, (n, 3))
    points = np.column_stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(4, 12, n)])
    poses = np.hstack([omegas, ts])
    x = np.array([tp.transform_project(poses[i], points[i]) for i in range(n)])
    A = np.array([tp.pose_jacobian(poses[i], points[i]) for i in range(n)])
    B = np.array([tp.point_jacobian(poses[i], points[i]) for i in range(n)])
    R = np.array([tp.exp_so3(poses[i, :3].copy()) for i in range(n)])
    # the three known answers of tests/test_transform_project.py:11-44
    return dict(poses=poses, points=points, x=x, A=A, B=B, R=R)


def main():
    install_stubs()
    os.makedirs(HERE, exist_ok=True)
    if "--only-vga-pyramid" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "dvo_vga_pyramid.npz"), **capture_vga_pyramid())
        return
    if "--only-pyramid" in sys.argv:
        pyr = capture_pyramid(120, 160, seed=4)
        for k in ("I0", "D0", "I1"):
            pyr.pop(k)
        np.savez_compressed(os.path.join(HERE, "dvo_pyramid.npz"), **pyr)
        return

    small = capture_dvo(48, 64, seed=3, weights_list=[None, "huber", "student-t", "tukey", "map"],
                        tag="s", keep_rows=True)
    np.savez_compressed(os.path.join(HERE, "dvo_small.npz"),
                        **{k: v for k, v in small.items()})

    # VGA: inputs are regenerated from the seed by the tests; only results kept
    vga = capture_dvo(480, 640, seed=0, weights_list=[None, "huber"], tag="v", keep_rows=False)
    for k in ("I0", "D0", "I1", "weight_map"):
        vga.pop(k)
    np.savez_compressed(os.path.join(HERE, "dvo_vga.npz"), **vga)

    pyr = capture_pyramid(120, 160, seed=4)
    for k in ("I0", "D0", "I1"):
        pyr.pop(k)
    np.savez_compressed(os.path.join(HERE, "dvo_pyramid.npz"), **pyr)

    np.savez_compressed(os.path.join(HERE, "dvo_vga_pyramid.npz"), **capture_vga_pyramid())

    np.savez_compressed(os.path.join(HERE, "pyref.npz"), **capture_pyref())

    tp = build_transform_project()
    np.savez_compressed(os.path.join(HERE, "ba_vectors.npz"), **capture_ba(tp))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
