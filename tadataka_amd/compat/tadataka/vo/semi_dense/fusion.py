from tadataka.vo.semi_dense._absent import absent

fusion = absent("fusion", "fusion")
