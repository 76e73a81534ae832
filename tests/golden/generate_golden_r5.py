#!/usr/bin/env python3
"""Round-5 fixture made by importing the reference's own pure-Python module in the build container:
  rigid_motion.npz   tadataka.rigid_motion.LeastSquaresRigidMotion(P, Q).solve() on 12 seeded point sets
                     (3-D and 2-D; exact similarity, noisy, a reflection-prone planar set)
Usage: python tests/golden/generate_golden_r5.py"""
import importlib.util
import os
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    spec = importlib.util.spec_from_file_location("ref_rigid_motion", os.path.join(REF, "tadataka", "rigid_motion.py"))
    rm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rm)
    out = {}
    rng = np.random.default_rng(2025)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # np.sum(generator) deprecation inside the reference
        for k in range(12):
            d = 3 if k < 9 else 2
            n = int(rng.integers(5, 60))
            P = rng.normal(size=(n, d)) * rng.uniform(0.5, 4.0)
            A = rng.normal(size=(d, d))
            Rt = np.linalg.svd(A)[0]
            if np.linalg.det(Rt) < 0:
                Rt[:, -1] *= -1
            s = rng.uniform(0.3, 3.0)
            t = rng.normal(size=d)
            Q = s * P @ Rt.T + t
            if k % 3 == 1:
                Q = Q + 0.05 * rng.normal(size=Q.shape)
            if k == 8:
                P[:, 2] = 0.0                       # planar
                Q = s * P @ Rt.T + t
            R, tt, ss = rm.LeastSquaresRigidMotion(P, Q).solve()
            out[f"P{k}"] = P; out[f"Q{k}"] = Q; out[f"R{k}"] = R; out[f"t{k}"] = tt; out[f"s{k}"] = np.float64(ss)
    out["n"] = np.array(12)
    np.savez_compressed(os.path.join(HERE, "rigid_motion.npz"), **out)
    print("rigid_motion.npz", os.path.getsize(os.path.join(HERE, "rigid_motion.npz")))


if __name__ == "__main__":
    main()
