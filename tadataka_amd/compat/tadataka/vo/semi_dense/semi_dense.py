from tadataka.vo.semi_dense._absent import absent

InvDepthEstimator = absent("semi_dense", "InvDepthEstimator")
InvDepthMapEstimator = absent("semi_dense", "InvDepthMapEstimator")
