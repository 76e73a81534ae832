from tadataka.vo.semi_dense._absent import absent

make_reference_selector = absent("reference", "make_reference_selector")
