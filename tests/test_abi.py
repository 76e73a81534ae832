"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/tadataka_hip.h declares; without a GPU the compute
entries fail loudly (there is no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import REPO, _has_gpu


def _header_symbols():
    hdr = open(os.path.join(REPO, "include", "tadataka_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(tdk_[a-z0-9_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(REPO, "tadataka_amd", "lib", "libtadataka_hip.so")):
        g.build()
    from tadataka_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    handle = lib.load()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in include/tadataka_hip.h but not exported"
    # and the ctypes prototypes cover the header exactly
    declared = set(syms) - {"tdk_version", "tdk_last_error"}
    assert declared == set(lib.PROTOTYPES)
    assert b"gfx950" in handle.tdk_version()


def test_no_product_code_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    use oracle/."""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "tadataka_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f), errors="replace").read()
                if re.search(r"(from|import)\s+oracle|liboracle|tdk_oracle|orc_", text):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu(lib):
    from tadataka_amd import ops
    assert lib.device_count() == 0
    with pytest.raises(lib.TdkError):
        lib.require_gpu()
    with pytest.raises(lib.TdkError):
        ops.warp_vecs(np.eye(4), np.zeros((2, 2)), np.ones(2))
    with pytest.raises(lib.TdkError):
        ops.DvoBatch(1, 8, 8)


def test_argument_validation_needs_no_gpu(lib):
    import ctypes as C
    h = lib.load()
    out = C.c_void_p()
    assert h.tdk_dvo_create(0, 8, 8, 1, 1.5, 0, C.byref(out)) == lib.TDK_ERR_INVALID_ARGUMENT
    assert h.tdk_dvo_create(1, 1, 8, 1, 1.5, 0, C.byref(out)) == lib.TDK_ERR_INVALID_ARGUMENT
    assert b"invalid argument" in h.tdk_last_error()
    d = C.c_double()
    T = (C.c_double * 16)(*np.array([[1., 0., 0., 2.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]).ravel())
    x0 = (C.c_double * 2)(0.1, 0.2); x1 = (C.c_double * 2)(0.3, 0.2)
    # calc_depth0 is scalar host arithmetic inside the library
    assert h.tdk_calc_depth0(T, x0, x1, C.byref(d)) == 0
    from oracle import oracle as orc
    assert d.value == orc.calc_depth0(np.array(T).reshape(4, 4), [0.1, 0.2], [0.3, 0.2])


def test_round5_entries_validate_their_arguments_without_a_gpu(lib):
    """tdk_rescale_skimage / tdk_set_option / tdk_ba_create_ex reject bad arguments before they touch a device."""
    import ctypes as C
    h = lib.load()
    img = np.zeros((4, 5))
    out = np.zeros((2, 3))
    m = np.array([2.0, 0.5, 2.0, 0.5])
    w = np.ones(2 * 65 + 1)
    dp = C.POINTER(C.c_double)
    p = lambda a: a.ctypes.data_as(dp)      # noqa: E731
    # kernel radius beyond the 64 the kernels hold
    assert h.tdk_rescale_skimage(p(img), 4, 5, p(out), 2, 3, p(m), p(w), 65, None, 0, 1) == lib.TDK_ERR_INVALID_ARGUMENT
    # a map whose scale is not positive
    bad = np.array([0.0, 0.5, 2.0, 0.5])
    assert h.tdk_rescale_skimage(p(img), 4, 5, p(out), 2, 3, p(bad), None, 0, None, 0, 1) == lib.TDK_ERR_INVALID_ARGUMENT
    assert h.tdk_rescale_skimage(None, 4, 5, p(out), 2, 3, p(m), None, 0, None, 0, 1) == lib.TDK_ERR_INVALID_ARGUMENT
    assert h.tdk_set_option(7, 1) == lib.TDK_ERR_INVALID_ARGUMENT
    assert h.tdk_set_option(0, 9) == lib.TDK_ERR_INVALID_ARGUMENT
    assert h.tdk_set_option(0, 1) == 0 and h.tdk_set_option(1, 1) == 0
    vp = np.zeros(3, dtype=np.int64); pt = np.arange(3, dtype=np.int64); xt = np.zeros((3, 2))
    hd = C.c_void_p()
    i64p = C.POINTER(C.c_int64)
    assert h.tdk_ba_create_ex(1, 3, vp.ctypes.data_as(i64p), pt.ctypes.data_as(i64p), p(xt), 3, 16, C.byref(hd)) \
        == lib.TDK_ERR_INVALID_ARGUMENT
    assert b"option" in h.tdk_last_error()
    n = C.c_int(5)
    assert h.tdk_debug_check_canaries(C.byref(n)) == 0 and n.value in (-1, 0)   # off (or on with nothing allocated)


def test_streaming_pyramid_kernel_keeps_its_prefetch_registers_untouched():
    """k_pyramid_stream issues its row prefetch and the identity level's stores behind the compiler's back and waits
    for the prefetch with its own s_waitcnt (csrc/pyramid.hip: stream_load); nothing may touch the destination
    registers in between.  The compiler cannot know, so the generated gfx950 assembly is checked (hipcc -S, no GPU)."""
    import importlib.util
    import shutil
    if shutil.which("/opt/rocm/bin/hipcc") is None:
        pytest.skip("no hipcc")
    spec = importlib.util.spec_from_file_location(
        "check_pyramid_isa", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                                          "check_pyramid_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernels, problems = mod.check(mod.assembly())
    assert kernels == 2 and not problems, problems


def test_the_isa_check_itself_sees_a_touched_prefetch_register(tmp_path):
    """tools/check_pyramid_isa.py on hand-made assembly: a copy out of a prefetch register before the kernel's wait is
    reported (this is what the register allocator did once, with two wait statements and a phi between them), the
    same instruction after the wait is not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_pyramid_isa", os.path.join(REPO, "tools", "check_pyramid_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    head = "_ZN1_16k_pyramid_streamILi1ELi3EEEv: ; @x\n"
    load = "\t;;#ASMSTART\n\tglobal_load_dwordx2 v[76:77], v[0:1], off\n\t;;#ASMEND\n"
    wait = ("\t;;#ASMSTART\n\ts_cmp_lg_u32 s0, 0\n\ts_cbranch_scc0 1f\n\ts_waitcnt vmcnt(8)\n\ts_branch 2f\n1:\n"
            "\ts_waitcnt vmcnt(0)\n2:\n\t;;#ASMEND\n")
    copy = "\tv_mov_b64_e32 v[2:3], v[76:77]\n"
    other = "\tv_add_f64 v[4:5], v[6:7], v[8:9]\n"
    good = tmp_path / "good.s"
    good.write_text(head + load + other + wait + copy + "\ts_endpgm\n")
    bad = tmp_path / "bad.s"
    bad.write_text(head + load + copy + wait + "\ts_endpgm\n")
    kernels, problems = mod.check(str(good))
    assert kernels == 1 and not problems, problems
    kernels, problems = mod.check(str(bad))
    assert kernels == 1 and len(problems) == 1 and "v_mov_b64_e32" in problems[0], problems
    stale = tmp_path / "stale.s"
    stale.write_text(head + other + "\ts_endpgm\n")
    assert mod.check(str(stale))[1], "a kernel without the prefetch / wait pattern must be reported (stale check)"
