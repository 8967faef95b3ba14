"""ctypes binding of the gfx950 rasterizer library (C ABI: include/gs_raster.h).

The library is built ahead of time by `make -C easy_gaussian_splatting_amd/csrc`
(`__graft_entry__.build()` does that) and lives IN-TREE next to this file.  There is no
fallback: if the shared object is missing or a stage fails, the call raises.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree build.  GS_LIB_PATH names another build of the same library -- a diagnostic VARIANT
# (`build_variant`: -DGS_BWD_CHECK, -DGS_BWD_ACC64, -DGS_EXACT_MATH, -DGS_CLOCK_PROBE, tuning constants) -- and is honoured only
# together with GS_ALLOW_VARIANT=1: every library names the flags it was built with (`gs_build_flags`), and one whose flags
# are not empty is refused otherwise, so that a stray environment variable cannot put a timing or checking build behind
# the numbers and the parity results.
LIB_PATH = os.environ.get("GS_LIB_PATH") or os.path.join(_HERE, "libgsraster.so")
CSRC_DIR = os.path.join(_HERE, "csrc")
VARIANT_DIR = os.path.join(os.path.dirname(_HERE), "build", "variants")   # never the package directory

GS_TILE = 16
GS_BUCKET = 64
GS_UNIT = 32
GS_REC_FLOATS = 12
GS_ROW_FLOATS = 12
GS_ROUND_BASE, GS_ROUND_SPLIT, GS_ROUND_LIVE, GS_ROUND_FRONT_N, GS_ROUND_LISTED_ALL, GS_ROUND_WORDS = 0, 1, 2, 3, 4, 8   # words of a depth-rounds block

_lib: Optional[ct.CDLL] = None
_lock = threading.Lock()

_P = ct.c_void_p
_I = ct.c_int
_L = ct.c_int64
_F = ct.c_float
_Z = ct.c_size_t

# name -> (restype, argtypes); must list every symbol include/gs_raster.h declares
SIGNATURES = {
    "gs_version": (_I, []),
    "gs_build_flags": (ct.c_char_p, []),
    "gs_last_error": (ct.c_char_p, []),
    "gs_arch": (ct.c_char_p, []),
    "gs_bin_groups": (_I, [_L]),
    "gs_bin_workspace_bytes": (_Z, [_I, _L, _I, _I]),
    "gs_project_fwd": (_I, [_P, _I, _L, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _F, _F, _F, _F, _I, _I, _I,
                            _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gs_bin_count": (_I, [_P, _I, _L, _I, _I, _P, _P, _Z, _P, _P, _P, _P, _P]),
    "gs_bin_emit_sort": (_I, [_P, _I, _L, _I, _I, _P, _P, _P, _Z, _P, _L, _L, _P, _P, _P, _P, _P, _P]),
    "gs_bins_workspace_bytes": (_Z, [_I, _L, _I, _I, _I, _L]),
    "gs_bins_count": (_I, [_P, _I, _L, _I, _I, _I, _P, _P, _P, _Z, _P, _L, _L, _P, _P, _P, _P, _P, _P]),
    "gs_bins_lists": (_I, [_P, _I, _L, _I, _I, _I, _P, _P, _Z, _P, _L, _P, _P, _P, _P, _P, _P]),
    "gs_walk_state_ints": (_Z, [_L]),
    "gs_blend_fwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P]),
    "gs_blend_bwd": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gs_unit_classes_ints": (_Z, [_L, _I, _I, _I]),
    "gs_project_bwd": (_I, [_P, _I, _L, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _F, _F, _F,
                            _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "gs_row_sums": (_I, [_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "gs_sh_adam_views": (_I, [_P, _I, _L, _I, _I, _P, _P, _L, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _L, _F, _P]),
    "gs_sh_grad_views": (_I, [_P, _I, _L, _I, _I, _P, _P, _P, _P, _P]),
    "gs_loss_workspace_floats": (_Z, [_I, _I]),
    "gs_l1_ssim_fwd": (_I, [_P, _I, _I, _F, _P, _P, _P, _I, _P, _P]),
    "gs_l1_ssim_bwd": (_I, [_P, _I, _I, _F, _P, _P, _P, _I, _P, _P, _P]),
    "gs_l1_ssim_fwd_slots": (_I, [_P, _I, _I, _F, _P, _P, _I, _I, _P, _P]),
    "gs_l1_ssim_bwd_slots": (_I, [_P, _I, _I, _F, _P, _P, _I, _P, _P, _P]),
    "gs_clamp01": (_I, [_P, _L, _P, _P, _P]),
    "gs_pack_view_step": (_I, [_P, _L, _F, _P, _P, _P, _P, _P, _P, _P]),
    "gs_update_statistics": (_I, [_P, _L, _F, _P, _P, _P, _P, _P]),
    "gs_guard_set": (_I, [_P, _L, _L]),
    "gs_guard_set_call": (_I, [_P, _L, _L]),
    "gs_info_mirror_set": (_I, [_P]),
    "gs_walk_mirror_set": (_I, [_P]),
    "gs_rounds_set": (_I, [_P, _P, _P, _P, _I]),
    "gs_round_split": (_I, [_P, _L, _P, _P, _F, _P, _P]),
    "gs_round_footprints": (_I, [_P, _L, _I, _I, _P, _P, _P, _P]),
    "gs_round_status": (_I, [_P, _P, _P]),
    "gs_step_status": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "gs_guard_flag_out": (_I, [_P, _P, _P, _P]),
    "gs_guard_merge": (_I, [_P, _P, _P, _I, _L]),
    "gs_step_applied": (_I, [_P, _P, _P]),
    "gs_adam_hyper": (_I, [_P, _I, _P, _F, _F, _L, _P]),
    "gs_step_inputs": (_I, [_P, _I, _P, _F, _F, _L, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gs_adam_step_dev": (_I, [_P, _L, _P, _P, _P, _I, _P, _P, _P, _F, _F, _F, _F, _P, _P]),
    "gs_scan_rows_workspace_ints": (_Z, [_I, _L]),
    "gs_scan_rows_i32": (_I, [_P, _I, _L, _P, _P, _P]),
    "gs_refine_flags": (_I, [_P, _L, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    "gs_refine_apply": (_I, [_P, _L, _I, _I, _P, _P, _L, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gs_project_bwd_adam": (_I, [_P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P, _P,
                                  _P, _P, _P, _P]),
    "gs_workspace_query": (_I, [_I, _L, _I, _I, _L, _L, _L, _L, _I, _I, _P, _P]),
    "gs_workspace_bind": (_I, [_P, _P, _L, _P, _L, _P, _L, _P, _P]),
    "gs_adam_step": (_I, [_P, _L, _P, _P, _P, _I, _P, _P, _P, _P, _F, _F, _F, _L, _F]),
    "gs_adam_step_stats": (_I, [_P, _L, _P, _P, _P, _I, _P, _P, _P, _P, _F, _F, _F, _L, _F, _L, _P, _P, _P, _P]),
}


class NativeLibraryError(RuntimeError):
    pass


def build(verbose: bool = False, clean: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU).  `clean`: from scratch -- no object of an
    earlier build (another revision, other flags) can end up in the library."""
    if clean:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], capture_output=True, text=True)
    proc = subprocess.run(["make", "-C", CSRC_DIR, "-j4"], capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout)
        print(proc.stderr)
    if proc.returncode != 0:
        raise NativeLibraryError("building libgsraster.so failed:\n" + proc.stderr[-4000:])
    return LIB_PATH


def build_variant(name: str, extra_flags: str) -> str:
    """A diagnostic variant of the library, `build/variants/libgsraster_<name>.so` (outside the package), compiled with
    additional flags from a scratch copy of the sources (so the product's objects are left alone).  It reports those flags
    through `gs_build_flags`; a process loads it through GS_LIB_PATH + GS_ALLOW_VARIANT=1 (tests/test_gpu_contributors.py does,
    in a child process); never loaded by the package on its own."""
    import glob
    import shutil
    import tempfile
    os.makedirs(VARIANT_DIR, exist_ok=True)
    out = os.path.join(VARIANT_DIR, f"libgsraster_{name}.so")
    srcs = [f for pat in ("*.hip", "*.h", "*.inc", "Makefile") for f in glob.glob(os.path.join(CSRC_DIR, pat))]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(f) for f in srcs + [os.path.join(_HERE, "..", "include", "gs_raster.h")]):
        return out
    d = tempfile.mkdtemp(prefix=f"gsvar_{name}_")
    try:
        header = os.path.abspath(os.path.join(_HERE, "..", "include", "gs_raster.h"))
        for f in srcs:
            text = open(f).read().replace("../../include/gs_raster.h", header)
            open(os.path.join(d, os.path.basename(f)), "w").write(text)
        proc = subprocess.run(["make", "-C", d, "-j4", f"EXTRA={extra_flags}", f"LIB={out}"], capture_output=True, text=True)
        if proc.returncode != 0:
            raise NativeLibraryError(f"building {out} failed:\n" + proc.stderr[-4000:])
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def lib() -> ct.CDLL:
    """Load (once) and return the native library; raises if it is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise NativeLibraryError(
                        f"{LIB_PATH} is missing: the HIP rasterizer has not been built. Run "
                        "`python -c 'import __graft_entry__ as g; g.build()'` or "
                        f"`make -C {CSRC_DIR}`. There is no CPU fallback.")
                L = ct.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)  # AttributeError => header/library mismatch
                    fn.restype = res
                    fn.argtypes = args
                flags = L.gs_build_flags().decode("utf-8", "replace")
                if flags and os.environ.get("GS_ALLOW_VARIANT") != "1":
                    raise NativeLibraryError(
                        f"{LIB_PATH} is a diagnostic variant of the rasterizer library (built with `{flags}`), not the product "
                        "build: refusing to run on it.  Unset GS_LIB_PATH, or set GS_ALLOW_VARIANT=1 if that is what you want.")
                _lib = L
    return _lib


def build_flags() -> str:
    """The extra preprocessor flags the loaded library was built with ("" = the product build)."""
    return lib().gs_build_flags().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().gs_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise NativeLibraryError(f"{what} failed (code {rc}): {msg}")
