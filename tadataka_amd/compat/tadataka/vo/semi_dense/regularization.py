from tadataka.vo.semi_dense._absent import absent

regularize = absent("regularization", "regularize")
