"""tadataka.vo.semi_dense.fusion (imported by examples/semi_dense_vo.py:15, called
at :85): Gaussian fusion of two hypothesis maps on the MI355X through
tdk_fusion_arrays (src/semi_dense/fusion.rs:3-42)."""
from tadataka.vo.semi_dense.hypothesis import HypothesisMap
from tadataka_amd import ops


def fusion(hypothesis1, hypothesis2):
    mu, var = ops.fusion_arrays(hypothesis1.inv_depth_map, hypothesis2.inv_depth_map,
                                hypothesis1.variance_map, hypothesis2.variance_map)
    return HypothesisMap(mu, var)
