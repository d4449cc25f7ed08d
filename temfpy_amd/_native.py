"""ctypes binding of ``libtemfpy_hip.so`` (C ABI declared in ``include/temfpy_hip.h``).

There is no CPU fallback: if the library is missing or a symbol is absent, ``load()``
raises, and every entry point raises on a non-zero status with the library's message.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_LIB = None
_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtemfpy_hip.so")

TMF_F64, TMF_C128 = 0, 1

# numpy mirrors of the descriptor structs (field order/offsets as in the header)
gemm_desc = np.dtype([("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("M", "<i4"), ("N", "<i4"), ("K", "<i4"),
                      ("lda", "<i4"), ("ldb", "<i4"), ("ldc", "<i4")])
panel_desc = np.dtype([("A", "<u8"), ("norms", "<u8"), ("n", "<i4"), ("w", "<i4"), ("lda", "<i4"), ("pad", "<i4")])
nested_desc = np.dtype([("C", "<u8"), ("Omega", "<u8"), ("dest", "<u8"), ("ncol", "<u8"), ("ld", "<u8"), ("D", "<i4"),
                        ("ldc", "<i4"), ("ldo", "<i4"), ("suffix", "<i4"), ("rows_ge", "<i4"), ("x_lo", "<i4"),
                        ("x_hi", "<i4"), ("maxc", "<i4")])
assert nested_desc.itemsize == 72
recon_desc = np.dtype([("T", "<u8"), ("X", "<u8"), ("Y", "<u8"), ("w", "<u8"), ("out", "<u8"), ("rows", "<i4"),
                       ("cols", "<i4"), ("q", "<i4"), ("inner", "<i4"), ("ldt", "<i4"), ("ldx", "<i4"), ("ldy", "<i4"),
                       ("mode", "<i4"), ("y_reverse", "<i4"), ("pad", "<i4")])
assert recon_desc.itemsize == 80
bcgs_desc = np.dtype([("base", "<u8"), ("scratch", "<u8"), ("norms", "<u8"), ("rows", "<i4"), ("ld", "<i4"),
                      ("c_begin", "<i4"), ("c_end", "<i4")])
assert bcgs_desc.itemsize == 40
det_site = np.dtype([("S", "<u8"), ("scale", "<u8"), ("idx_base", "<u8"), ("out_base", "<u8"), ("lds", "<i4"), ("pad", "<i4")])
assert det_site.itemsize == 40
norms_desc = np.dtype([("src", "<u8"), ("out", "<u8"), ("n", "<i4"), ("c", "<i4"), ("lds_", "<i4"), ("pad", "<i4")])
jacobi_desc = np.dtype([("X", "<u8"), ("V", "<u8"), ("U", "<u8"), ("s", "<u8"), ("count", "<u8"),
                        ("thresh2", "<f8"), ("p", "<i4"), ("ldx", "<i4"), ("ldv", "<i4"), ("ldu", "<i4")])
schur_desc = np.dtype([("W", "<u8"), ("S", "<u8"), ("det", "<u8"), ("mb", "<i4"), ("mk", "<i4"), ("k", "<i4"),
                       ("ldw", "<i4"), ("lds", "<i4"), ("pad", "<i4")])
lublock_desc = np.dtype([("W", "<u8"), ("det", "<u8"), ("piv", "<u8"), ("T", "<u8"), ("mb", "<i4"), ("mk", "<i4"),
                         ("k", "<i4"), ("ldw", "<i4")])
assert lublock_desc.itemsize == 48
diaginv_desc = np.dtype([("W", "<u8"), ("det", "<u8"), ("inv", "<u8"), ("mb", "<i4"), ("mk", "<i4"), ("k", "<i4"), ("ldw", "<i4")])
assert diaginv_desc.itemsize == 40
det_desc = np.dtype([("S", "<u8"), ("scale", "<u8"), ("bra_idx", "<u8"), ("ket_idx", "<u8"), ("out", "<u8"),
                     ("sb", "<i4"), ("sk", "<i4"), ("lds", "<i4"), ("n", "<i4"), ("nsb", "<i4"), ("nsk", "<i4"),
                     ("a0", "<i4"), ("a1", "<i4")])
gather_desc = np.dtype([("src", "<u8"), ("dst", "<u8"), ("row_sel", "<u8"), ("col_sel", "<u8"),
                        ("row_sign", "<u8"), ("col_sign", "<u8"), ("phys", "<u8"), ("rows", "<i4"),
                        ("cols", "<i4"), ("lds_", "<i4"), ("ldd", "<i4"), ("ldp", "<i4"), ("pad", "<i4")])
rescale_desc = np.dtype([("A", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("ld", "<i4"), ("pad", "<i4")])
gauge_desc = np.dtype([("V", "<u8"), ("start", "<u8"), ("n", "<i4"), ("k", "<i4"), ("ld", "<i4"), ("from_top", "<i4")])
colnorm_desc = np.dtype([("src", "<u8"), ("dst", "<u8"), ("n", "<i4"), ("c", "<i4"), ("lds_", "<i4"),
                         ("ldd", "<i4"), ("reverse", "<i4"), ("flip_odd", "<i4")])
site_in = np.dtype([("mode", "<i4"), ("k_b", "<i4"), ("nf_b", "<i4"), ("chi_b", "<i4"), ("k_k", "<i4"),
                    ("nf_k", "<i4"), ("chi_k", "<i4"), ("pad", "<i4")])
site_out = np.dtype([("mb", "<i4"), ("mk", "<i4"), ("k_always", "<i4"), ("sb", "<i4"), ("sk", "<i4"),
                     ("n_sectors", "<i4"), ("idx_bytes", "<i8"), ("out_elems", "<i8")])
sector = np.dtype([("q", "<i4"), ("r0", "<i4"), ("r1", "<i4"), ("c0", "<i4"), ("c1", "<i4"), ("n", "<i4"),
                   ("bra_off", "<i8"), ("ket_off", "<i8"), ("out_off", "<i8")])

pf_desc = np.dtype([("N", "<u8"), ("scale", "<u8"), ("bra_idx", "<u8"), ("ket_idx", "<u8"), ("out", "<u8"),
                    ("nn", "<i4"), ("ldn", "<i4"), ("n1", "<i4"), ("n2", "<i4"), ("nsb", "<i4"), ("nsk", "<i4"),
                    ("a0", "<i4"), ("a1", "<i4")])
assert pf_desc.itemsize == 72
nambu_asm_desc = np.dtype([("src", "<u8"), ("dst", "<u8"), ("col_src", "<u8"), ("col_conj", "<u8"), ("n2", "<i4"),
                           ("lds_", "<i4"), ("ldd", "<i4"), ("pad", "<i4")])
nambu_w_desc = np.dtype([("Vr", "<u8"), ("W", "<u8"), ("idx1", "<u8"), ("idx2", "<u8"), ("L", "<i4"), ("na", "<i4"),
                         ("nb", "<i4"), ("ldv", "<i4"), ("ldw", "<i4"), ("pad", "<i4")])
pf_matrix_desc = np.dtype([("S", "<u8"), ("N", "<u8"), ("na", "<i4"), ("nb", "<i4"), ("lds_", "<i4"), ("ldn", "<i4")])
assert nambu_asm_desc.itemsize == 48 and nambu_w_desc.itemsize == 56 and pf_matrix_desc.itemsize == 32
copy_desc = np.dtype([("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("lds_", "<i4"), ("ldd", "<i4"),
                      ("flags", "<i4"), ("pad", "<i4")])
assert copy_desc.itemsize == 40
qr_desc = np.dtype([("A", "<u8"), ("R", "<u8"), ("m", "<i4"), ("n", "<i4"), ("lda", "<i4"), ("ldr", "<i4"),
                    ("flags", "<i4"), ("pad", "<i4")])
assert qr_desc.itemsize == 40
slab_desc = np.dtype([("A", "<u8"), ("Q", "<u8"), ("R", "<u8"), ("n", "<i4"), ("c", "<i4"), ("lda", "<i4"), ("ldq", "<i4"),
                      ("ldr", "<i4"), ("flags", "<i4")])
assert slab_desc.itemsize == 48
site_job = np.dtype([("mode", "<i4"), ("cut_b", "<i4"), ("cut_k", "<i4"), ("k_b", "<i4"), ("nf_b", "<i4"),
                     ("k_k", "<i4"), ("nf_k", "<i4"), ("sec_cap", "<i4"), ("row_off", "<i8"), ("col_off", "<i8"),
                     ("bra_off", "<i8"), ("sec_off", "<i8"), ("idx_off", "<i8"), ("idx_cap", "<i8")])
assert site_job.itemsize == 80
assert gemm_desc.itemsize == 48 and panel_desc.itemsize == 32 and norms_desc.itemsize == 32 and jacobi_desc.itemsize == 64
assert schur_desc.itemsize == 48 and det_desc.itemsize == 72 and gather_desc.itemsize == 80
assert colnorm_desc.itemsize == 40 and sector.itemsize == 48 and site_out.itemsize == 40

SYMBOLS = [
    "tmf_last_error", "tmf_version", "tmf_device_count", "tmf_gemm_batched", "tmf_gemm_tall_batched", "tmf_orth_panel_batched",
    "tmf_bcgs_work_bytes", "tmf_bcgs_batched", "tmf_jacobi_batched", "tmf_svd_left_batched",
    "tmf_jacobi_block_batched", "tmf_nested_products_batched", "tmf_recon_error_batched", "tmf_lu_schur_batched",
    "tmf_det_gather_batched", "tmf_det_reduced_batched", "tmf_det_ppt_batched", "tmf_det_ppt_batched_w", "tmf_det_ppt_stamps", "tmf_block_orth_batched", "tmf_house_slab_stamps", "tmf_pfaffian_sweep", "tmf_pf_result_dims", "tmf_pf_result_bond", "tmf_pf_result_site",
    "tmf_pf_result_block", "tmf_pf_result_checks", "tmf_pf_result_flat", "tmf_pf_result_download", "tmf_pf_result_free", "tmf_onishi_norms", "tmf_host_parallel_for", "tmf_gemm_set_4m", "tmf_block_orth_stamps", "tmf_transpose", "tmf_fill_normal",
    "tmf_gather_signed_batched", "tmf_normalise_columns_batched", "tmf_canonical_gauge_batched", "tmf_rescale_pow2_batched", "tmf_column_norms_batched", "tmf_cut_vectors",
    "tmf_site_prepare", "tmf_cut_vectors_batch", "tmf_site_prepare_batch", "tmf_det_tiles_build", "tmf_pf_gather_batched",
    "tmf_nambu_assemble_batched", "tmf_nambu_w_batched", "tmf_pf_matrix_batched", "tmf_copy_blocks_batched", "tmf_house_qr_batched", "tmf_jacobi_compact_batched", "tmf_house_slab_batched", "tmf_house_qr_regs_batched", "tmf_house_form_q_batched",
    "tmf_host_register", "tmf_host_unregister", "tmf_memcpy_async", "tmf_lu_block_batched", "tmf_lu_trsm_batched", "tmf_diag_inverse_batched", "tmf_diag_inverse_verdict", "tmf_launch_condition", "tmf_export_words",
    "tmf_ctx_create", "tmf_ctx_destroy", "tmf_sweep_begin", "tmf_sweep_entangled", "tmf_sweep_sites", "tmf_sweep_download",
    "tmf_sweep_query", "tmf_sweep_wait", "tmf_sweep_info_get", "tmf_sweep_stage_name", "tmf_sweep_device_out",
    "tmf_slater_sweep", "tmf_result_dims", "tmf_result_bond", "tmf_result_site", "tmf_result_block", "tmf_result_checks",
    "tmf_result_free",
]

# ---- sweep-level structs (include/temfpy_hip.h, "Sweep level") ---------------------------------------------
SWEEP_CHECKS, SWEEP_TIME_KERNELS, SWEEP_RANGE_BCGS, SWEEP_NO_CHOLQR = 1, 2, 4, 8
SWEEP_DET_REDUCED, SWEEP_DET_DIRECT, SWEEP_C_ON_DEVICE, SWEEP_TWO_PASSES, SWEEP_LU_SINGLE, SWEEP_NARROW_BCGS, SWEEP_ONE_STREAM = (
    16, 32, 64, 128, 256, 512, 1024)
SWEEP_LU_PIVOTED, SWEEP_LU_FORCE_FALLBACK, SWEEP_UNFUSED_BCGS = 2048, 4096, 8192


class SweepParams(C.Structure):
    _fields_ = [("L", C.c_int64), ("chi_max", C.c_int64), ("svd_min", C.c_double), ("degeneracy_tol", C.c_double),
                ("sectors", C.c_void_p), ("ortho_center", C.c_int64), ("site_lo", C.c_int64), ("site_hi", C.c_int64),
                ("n_sectors", C.c_int32), ("is_complex", C.c_int32), ("host_threads", C.c_int32), ("flags", C.c_uint32)]


class SweepDims(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("ncut", "cap", "ns", "sec_tot", "bra_tot", "e_tot", "out_elems", "elem_bytes")]


SWEEP_ARRAYS = ("my_cuts", "c_sets", "c_lam", "c_q", "c_chi", "c_chk", "e_pool", "e_off", "kk_cut", "nfl", "nfr", "mode",
                "sec_off", "nsec", "sectors", "out_off", "bra_off", "chi_b", "chi_k", "bra_p", "bra_alpha", "det", "out")


class SweepPtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in SWEEP_ARRAYS]


class SweepInfo(C.Structure):
    _fields_ = [("stage_ms", C.c_double * 16), ("gemm_ms", C.c_double), ("gemm_flops", C.c_double), ("det_ms", C.c_double),
                ("det_flops", C.c_double), ("det_all_ms", C.c_double), ("n_det", C.c_int64), ("n_gemm_launches", C.c_int64),
                ("det_kind", C.c_int32), ("det_order", C.c_int32), ("range_width", C.c_int32), ("range_iterations", C.c_int32),
                ("range_floor", C.c_double), ("n_fermion", C.c_int64), ("device_bytes", C.c_int64),
                ("lu_min_pivot", C.c_double), ("lu_max_inverse", C.c_double), ("lu_fallbacks", C.c_int64),
                ("gemm_split_ms", C.c_double * 3), ("gemm_split_flops", C.c_double * 3), ("gemm_split_launches", C.c_int64 * 3)]


class BondView(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("x", "chi", "k", "n_filled_left", "n_filled_right", "n_checked")] + \
               [("e", C.c_void_p), ("masks", C.c_void_p), ("lam_raw", C.c_void_p), ("q_left", C.c_void_p)]


class SiteView(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("site", "chi_bra", "chi_ket", "n_blocks")] + \
               [("mode", C.c_int32), ("pad", C.c_int32), ("det_always", C.c_double * 2), ("bra_p", C.c_void_p),
                ("bra_alpha", C.c_void_p)]


class BlockView(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("q", "r0", "r1", "c0", "c1", "n")] + [("data", C.c_void_p)]


assert C.sizeof(SweepParams) == 80 and C.sizeof(SweepDims) == 64 and C.sizeof(SweepPtrs) == 23 * 8


class PfBondView(C.Structure):
    _fields_ = [("x", C.c_int64), ("k", C.c_int32), ("chi", C.c_int32), ("p_left", C.c_int32), ("p_right", C.c_int32),
                ("e", C.c_void_p), ("sets", C.c_void_p), ("lam_raw", C.c_void_p)]


class PfSiteView(C.Structure):
    _fields_ = [("site", C.c_int64), ("mode", C.c_int32), ("qtotal", C.c_int32), ("chi_bra", C.c_int32), ("chi_ket", C.c_int32),
                ("n_blocks", C.c_int32), ("pad", C.c_int32), ("norm", C.c_double), ("leg_idx_bra", C.c_void_p)]


class PfBlockView(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_bra", "n_ket", "r0", "r1", "c0", "c1")] + [("data", C.c_void_p)]


class PfFlat(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_bonds", "n_sites", "n_blocks", "out_elems")] + \
               [(n, C.c_void_p) for n in ("bond", "e_off", "e", "sets_off", "sets", "lam_off", "lam_raw", "site", "norm", "leg_off",
                                          "leg_idx_bra", "blk_off", "blk", "out")]


E_HALF_MODES = -4          # tmf_pfaffian_sweep: eigenvalue-1/2 modes at a cut, take the Python driver


def sweep_spec(dims, cplx, want_out=True):
    """{name: (dtype, shape)} of the result arrays of a sweep (the destinations of tmf_sweep_download)."""
    cdt = np.complex128 if cplx else np.float64
    nc, cap, ns = int(dims.ncut), int(dims.cap), int(dims.ns)
    return {"my_cuts": (np.int64, (nc,)), "c_sets": (np.uint64, (nc, cap, 2)), "c_lam": (np.float64, (nc, cap)),
            "c_q": (np.int32, (nc, cap)), "c_chi": (np.int64, (nc,)), "c_chk": (np.int64, (nc,)),
            "e_pool": (np.float64, (int(dims.e_tot) + 1,)), "e_off": (np.int64, (nc,)), "kk_cut": (np.int32, (nc,)),
            "nfl": (np.int32, (nc,)), "nfr": (np.int32, (nc,)), "mode": (np.int32, (ns,)), "sec_off": (np.int64, (ns,)),
            "nsec": (np.int64, (ns,)), "sectors": (sector, (int(dims.sec_tot) + 1,)), "out_off": (np.int64, (ns,)),
            "bra_off": (np.int64, (ns,)), "chi_b": (np.int64, (ns,)), "chi_k": (np.int64, (ns,)),
            "bra_p": (np.int32, (int(dims.bra_tot) + 1,)), "bra_alpha": (np.int32, (int(dims.bra_tot) + 1,)),
            "det": (cdt, (ns,)), "out": (cdt, (int(dims.out_elems) if want_out else 0,))}



class NativeError(RuntimeError):
    pass


def load():
    """dlopen the library and check that every symbol of the header is exported."""
    global _LIB
    if _LIB is not None:
        return _LIB
    host_only = os.environ.get("TMF_ASAN_LIB")      # tools/run_host_asan.sh: sanitizer build of the host entry points
    if host_only:
        lib = C.CDLL(host_only)
        lib.tmf_last_error.restype = C.c_char_p
        _set_host_argtypes(lib)
        _LIB = lib
        return lib
    # PyTorch ships its own HIP runtime.  Whichever of the two copies is loaded first serves the process; loaded second,
    # /opt/rocm's reported "no ROCm-capable device" to this library (build() followed by smoke() in ONE process).  Every
    # other entry path imports torch first - do the same here.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(_PATH):
        raise NativeError(
            f"{_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    lib = C.CDLL(_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise NativeError(f"{_PATH} does not export {s}")
    lib.tmf_last_error.restype = C.c_char_p
    vp, i32, i64, f64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_uint64
    lib.tmf_gemm_batched.argtypes = [i32, i32, f64, f64, vp, vp, i32, i32, vp]
    lib.tmf_gemm_tall_batched.argtypes = [i32, f64, f64, vp, vp, i32, vp]
    lib.tmf_orth_panel_batched.argtypes = [i32, vp, i32, i32, i32, vp]
    lib.tmf_jacobi_batched.argtypes = [i32, vp, i32, i32, vp, vp]
    lib.tmf_svd_left_batched.argtypes = [i32, vp, i32, i32, vp, vp]
    lib.tmf_jacobi_compact_batched.argtypes = [i32, vp, i32, i32, vp, vp]
    lib.tmf_nested_products_batched.argtypes = [i32, vp, i32, i32, i32, vp]
    lib.tmf_recon_error_batched.argtypes = [i32, vp, vp, i32, vp]
    lib.tmf_jacobi_block_batched.argtypes = [i32, i32, vp, i32, i32, vp, vp]
    lib.tmf_bcgs_work_bytes.argtypes = [vp, i32]
    lib.tmf_bcgs_work_bytes.restype = i64
    lib.tmf_bcgs_batched.argtypes = [i32, vp, vp, i32, i32, i32, vp, i64, vp]
    lib.tmf_lu_schur_batched.argtypes = [i32, vp, i32, i32, vp]
    lib.tmf_det_gather_batched.argtypes = [i32, i32, vp, i32, i32, vp]
    lib.tmf_det_ppt_batched.argtypes = [i32, vp, i32, i32, vp]
    lib.tmf_det_ppt_batched_w.argtypes = [i32, vp, i32, i32, i32, vp]
    lib.tmf_det_tiles_build.argtypes = [i32, vp, vp, vp, vp, i32, i64, vp, i64, vp, i64, vp, vp, vp, vp]
    lib.tmf_det_tiles_build.restype = i64
    lib.tmf_det_reduced_batched.argtypes = [i32, i32, vp, i32, i32, vp]
    lib.tmf_pf_gather_batched.argtypes = [i32, i32, vp, i32, i32, vp]
    for fn in (lib.tmf_nambu_assemble_batched, lib.tmf_nambu_w_batched, lib.tmf_pf_matrix_batched):
        fn.argtypes = [vp, i32, vp]
    lib.tmf_copy_blocks_batched.argtypes = [i32, vp, i32, i32, vp]
    lib.tmf_house_qr_batched.argtypes = [i32, vp, i32, i32, i32, vp]
    lib.tmf_house_slab_batched.argtypes = [i32, vp, i32, i32, i32, vp]
    lib.tmf_house_qr_regs_batched.argtypes = [i32, vp, i32, i32, i32, vp]
    lib.tmf_house_form_q_batched.argtypes = [i32, vp, i32, i32, i32, vp]
    lib.tmf_transpose.argtypes = [i32, vp, vp, i32, vp]
    lib.tmf_fill_normal.argtypes = [i32, vp, i64, u64, vp]
    lib.tmf_gather_signed_batched.argtypes = [i32, vp, i32, vp]
    lib.tmf_normalise_columns_batched.argtypes = [i32, vp, i32, vp]
    lib.tmf_canonical_gauge_batched.argtypes = [i32, vp, i32, i32, i32, vp]
    lib.tmf_rescale_pow2_batched.argtypes = [i32, vp, i32, vp, vp, vp]
    lib.tmf_column_norms_batched.argtypes = [i32, vp, i32, vp]
    _set_host_argtypes(lib)
    lib.tmf_lu_block_batched.argtypes = [i32, vp, i32, i32, i32, i32, vp]
    lib.tmf_lu_trsm_batched.argtypes = [i32, vp, i32, i32, i32, i32, vp]
    lib.tmf_diag_inverse_batched.argtypes = [i32, vp, i32, i32, vp, vp]
    lib.tmf_diag_inverse_verdict.argtypes = [vp, i32, C.c_double, i32, vp, vp, vp]
    lib.tmf_launch_condition.argtypes = [vp]
    lib.tmf_export_words.argtypes = [vp, vp, i64, vp]
    lib.tmf_launch_condition.restype = None
    lib.tmf_ctx_create.argtypes = [i32, C.POINTER(vp)]
    lib.tmf_ctx_destroy.argtypes = [vp]
    lib.tmf_ctx_destroy.restype = None
    lib.tmf_sweep_begin.argtypes = [vp, vp, C.POINTER(SweepParams)]
    lib.tmf_sweep_entangled.argtypes = [vp, i32, i32, C.POINTER(f64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(i64)]
    lib.tmf_sweep_sites.argtypes = [vp, C.POINTER(SweepDims)]
    lib.tmf_sweep_download.argtypes = [vp, C.POINTER(SweepPtrs), i32, C.POINTER(i64)]
    lib.tmf_sweep_query.argtypes = [vp, i64]
    lib.tmf_sweep_wait.argtypes = [vp, i64, C.POINTER(f64), C.POINTER(C.c_int32)]
    lib.tmf_sweep_info_get.argtypes = [vp, C.POINTER(SweepInfo)]
    lib.tmf_sweep_stage_name.argtypes = [i32]
    lib.tmf_sweep_stage_name.restype = C.c_char_p
    lib.tmf_sweep_device_out.argtypes = [vp, C.POINTER(u64), C.POINTER(i64)]
    lib.tmf_slater_sweep.argtypes = [vp, vp, C.POINTER(SweepParams), f64, C.POINTER(vp)]
    lib.tmf_result_dims.argtypes = [vp, C.POINTER(SweepDims), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    lib.tmf_result_bond.argtypes = [vp, i64, C.POINTER(BondView)]
    lib.tmf_result_site.argtypes = [vp, i64, C.POINTER(SiteView)]
    lib.tmf_result_block.argtypes = [vp, i64, i64, C.POINTER(BlockView)]
    lib.tmf_result_checks.argtypes = [vp, C.POINTER(f64), C.POINTER(C.c_int32)]
    lib.tmf_result_free.argtypes = [vp]
    lib.tmf_result_free.restype = None
    lib.tmf_pfaffian_sweep.argtypes = [vp, vp, C.POINTER(SweepParams), f64, C.POINTER(vp)]
    lib.tmf_pf_result_dims.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_int32), C.POINTER(SweepInfo)]
    lib.tmf_pf_result_bond.argtypes = [vp, i64, C.POINTER(PfBondView)]
    lib.tmf_pf_result_site.argtypes = [vp, i64, C.POINTER(PfSiteView)]
    lib.tmf_pf_result_block.argtypes = [vp, i64, i64, C.POINTER(PfBlockView)]
    lib.tmf_pf_result_checks.argtypes = [vp, vp, vp, vp]
    lib.tmf_pf_result_flat.argtypes = [vp, C.POINTER(PfFlat)]
    lib.tmf_pf_result_download.argtypes = [vp, vp]
    lib.tmf_pf_result_free.argtypes = [vp]
    lib.tmf_pf_result_free.restype = None
    lib.tmf_onishi_norms.argtypes = [vp, vp, i32, vp]
    lib.tmf_host_register.argtypes = [vp, i64]
    lib.tmf_host_unregister.argtypes = [vp]
    lib.tmf_memcpy_async.argtypes = [vp, vp, i64, i32, vp]
    _LIB = lib
    return lib


def _set_host_argtypes(lib):
    vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
    lib.tmf_cut_vectors.argtypes = [vp, i32, i32, i64, f64, f64, vp, i32, i64, vp, vp, vp, vp, vp]
    lib.tmf_site_prepare.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i64, vp]
    lib.tmf_cut_vectors_batch.argtypes = [i32, vp, vp, vp, vp, i64, f64, f64, vp, i32, i64, vp, vp, vp, vp, vp, i32]
    lib.tmf_site_prepare_batch.argtypes = [i32, vp, vp, vp, vp, i64] + [vp] * 9 + [i32]
    lib.tmf_det_tiles_build.argtypes = [i32, vp, vp, vp, vp, i32, i64, vp, i64, vp, i64, vp, vp, vp, vp]
    lib.tmf_det_tiles_build.restype = i64


def check(status: int, what: str):
    if status != 0:
        msg = load().tmf_last_error().decode()
        if status == -1:
            raise ValueError(f"{what}: {msg}")
        if status == -3:
            raise NotImplementedError(f"{what}: {msg}")
        raise NativeError(f"{what}: {msg} (status {status})")


JACOBI_SWEEP_CAP = 60   # csrc/jacobi.hip: a problem that reports this many sweeps still had rotations pending


def check_jacobi_sweeps(sweeps, what="Jacobi iteration"):
    """LAPACK's eigh / svd raise LinAlgError when they do not converge (numpy.linalg); so does this."""
    sweeps = np.asarray(sweeps)
    if sweeps.size and int(sweeps.max()) >= JACOBI_SWEEP_CAP:
        bad = int(np.count_nonzero(sweeps >= JACOBI_SWEEP_CAP))
        raise np.linalg.LinAlgError(f"{what} did not converge in {JACOBI_SWEEP_CAP} sweeps ({bad} of {sweeps.size} problems)")


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ---- host entry points (no GPU) ---------------------------------------------------------------
def cut_vectors(e, filled_left, chi_max, svd_min, degeneracy_tol, sectors=None):
    """schmidt_utils.lowest_sums + the ordering of slater.py:672-689.  Returns
    (sets uint64[chi,2], lam_raw[chi], q_left[chi], n_checked)."""
    lib = load()
    e = np.ascontiguousarray(e, np.float64)
    k = e.size
    cm = int(chi_max) if chi_max else 0
    sec = None if sectors is None else np.ascontiguousarray(sectors, np.int64)
    cap = cm + 1 if cm > 0 else 1024
    while True:
        sets = np.zeros((cap, 2), np.uint64)
        lam = np.zeros(cap)
        q = np.zeros(cap, np.int32)
        chi, nchk = C.c_int64(0), C.c_int64(0)
        st = lib.tmf_cut_vectors(_p(e), k, int(filled_left), cm, float(svd_min), float(degeneracy_tol),
                                 None if sec is None else _p(sec), 0 if sec is None else sec.size, cap, _p(sets),
                                 _p(lam), _p(q), C.byref(chi), C.byref(nchk))
        if st == -3 and chi.value > cap:
            cap = int(chi.value)
            continue
        check(st, "tmf_cut_vectors")
        n = chi.value
        return sets[:n], lam[:n], q[:n], nchk.value


def site_prepare(mode, k_b, nf_b, sets_b, q_b, k_k, nf_k, sets_k, q_k):
    """Integer part of MPSTensorData.from_schmidt_vectors / to_npc_array (slater.py:1023-1141)."""
    lib = load()
    chi_b, chi_k = len(sets_b), len(sets_k)
    sin = np.zeros(1, site_in)
    sin["mode"], sin["k_b"], sin["nf_b"], sin["chi_b"] = mode, k_b, nf_b, chi_b
    sin["k_k"], sin["nf_k"], sin["chi_k"] = k_k, nf_k, chi_k
    mb_cap, mk_cap = k_b + nf_b + 1, k_k + nf_k
    row_sel, row_sign = np.zeros(mb_cap, np.int32), np.zeros(mb_cap, np.int8)
    col_sel, col_sign = np.zeros(max(mk_cap, 1), np.int32), np.zeros(max(mk_cap, 1), np.int8)
    bra_p, bra_alpha = np.zeros(2 * chi_b, np.int32), np.zeros(2 * chi_b, np.int32)
    sec_cap = chi_k + 1
    secs = np.zeros(sec_cap, sector)
    idx_cap = (2 * chi_b + chi_k) * (max(k_b + 1, k_k) + 1) + 16
    pool = np.zeros(idx_cap, np.uint8)
    sout = np.zeros(1, site_out)
    sets_b = np.ascontiguousarray(sets_b, np.uint64)
    sets_k = np.ascontiguousarray(sets_k, np.uint64)
    q_b = np.ascontiguousarray(q_b, np.int32)
    q_k = np.ascontiguousarray(q_k, np.int32)
    st = lib.tmf_site_prepare(_p(sin), _p(sets_b), _p(q_b), _p(sets_k), _p(q_k), _p(row_sel), _p(row_sign),
                              _p(col_sel), _p(col_sign), _p(bra_p), _p(bra_alpha), _p(secs), sec_cap, _p(pool),
                              idx_cap, _p(sout))
    check(st, "tmf_site_prepare")
    o = sout[0]
    return dict(mb=int(o["mb"]), mk=int(o["mk"]), k=int(o["k_always"]), sb=int(o["sb"]), sk=int(o["sk"]),
                row_sel=row_sel[: o["mb"]], row_sign=row_sign[: o["mb"]], col_sel=col_sel[: o["mk"]],
                col_sign=col_sign[: o["mk"]], bra_p=bra_p, bra_alpha=bra_alpha, sectors=secs[: o["n_sectors"]].copy(),
                idx_pool=pool[: o["idx_bytes"]].copy(), out_elems=int(o["out_elems"]))


def reduced_det_lds(el, n, sb, sk, nsk, na):
    """Dynamic LDS bytes of one tmf_det_reduced_batched tile (layout: csrc/det_reduced.hip);
    works on scalars and NumPy arrays alike."""
    a16 = lambda x: (x + 15) & ~15  # noqa: E731
    return (a16(sb * sk * el) + a16(nsk * n) + a16(nsk * 8) + a16(na * n)
            + 4 * (((n | 1) * sk + 264) * el + 576) + 16)


def ppt_det_lds(el, sb, sk, nsk, na, n):
    """Dynamic LDS bytes of one tmf_det_ppt_batched tile (layout: csrc/det_ppt.hip)."""
    a16 = lambda x: (x + 15) & ~15  # noqa: E731
    return a16(a16(sb * sk * el) + (nsk + na) * 8) + 4 * (np.maximum(264, n * n) * el + 288) + a16(12 * na + 3 * nsk + 16)
