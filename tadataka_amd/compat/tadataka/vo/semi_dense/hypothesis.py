from tadataka.vo.semi_dense._absent import absent

HypothesisMap = absent("hypothesis", "HypothesisMap")
