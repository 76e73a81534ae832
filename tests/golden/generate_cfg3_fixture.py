#!/usr/bin/env python3
"""Generates tests/golden/semi_dense_cfg3.npz: the flag histogram and array
digests of BASELINE configs[2] as SURVEY.md section 8(d) defines it -- 640x480,
key/ref textures with baseline (0.1, 0, 0), Params(0.5, 10, 0.01, 0.01, 0.002,
0.02), age map Bernoulli(0.3) (seed 1), prior depth GT * U(0.9, 1.1), prior
variance 0.05 -- for increment_age, propagate and update_depth on those inputs.

The Rust crate cannot be compiled in the build container (no cargo), so the
values come from the CPU oracle (oracle/tdk_oracle.c, itself pinned to the Rust
unit-test literals by tests/test_oracle_literals.py): this fixture freezes WHAT
WORK the configuration is (how many pixels take which exit), so that the oracle
and the HIP path can both be held to it, and later changes to either show up.

Usage: python tests/golden/generate_cfg3_fixture.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from tadataka_amd import synthetic   # noqa: E402
from oracle import oracle as orc     # noqa: E402

PARAMS = (0.5, 10.0, 0.01, 0.01, 0.002, 0.02)
DEFAULTS = (1.0, 10.0, 0.01)      # default_depth, default_variance, uncertaintity_bias


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def compute(height=480, width=640, seed=1):
    c = synthetic.make_semi_dense_case(height, width, seed=seed)
    key = (c["cam"], c["key_image"], c["T_wk"])
    ref = (c["cam"], c["ref_image"], c["T_wr"])
    T10 = np.linalg.inv(c["T_wk"]) @ c["T_wr"]     # previous frame (= the reference frame) -> key frame
    p = orc.make_params(*PARAMS)
    age1 = orc.increment_age(c["age"], c["cam"], c["cam"], T10, c["prior_depth"])
    d1, v1 = orc.propagate(T10, c["cam"], c["cam"], c["prior_depth"], c["prior_variance"], *DEFAULTS)
    d, v, f = orc.update_depth(key, [ref], c["age"], c["prior_depth"], c["prior_variance"], p)
    hist = np.array([(f == -b).sum() for b in range(10)], dtype=np.int64)
    return dict(flag_histogram=hist, n_age1_nonzero=np.int64((age1 > 0).sum()),
                n_propagated=np.int64((d1 != DEFAULTS[0]).sum()),
                sha_age1=digest(age1), sha_depth1=digest(d1), sha_var1=digest(v1),
                sha_depth=digest(d), sha_var=digest(v), sha_flag=digest(f),
                params=np.array(PARAMS), defaults=np.array(DEFAULTS),
                shape=np.array([height, width]), seed=np.int64(seed))


if __name__ == "__main__":
    out = compute()
    np.savez_compressed(os.path.join(HERE, "semi_dense_cfg3.npz"), **out)
    print("flag histogram (0, -1, ..., -9):", out["flag_histogram"], "of", 480 * 640)
