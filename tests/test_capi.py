"""The C-ABI library: builds for gfx950, loads without a GPU, exports every symbol the header
declares; the Python front-end mirrors the reference interface's error behaviour."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from easy_gaussian_splatting_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    return _native


def test_header_symbols_all_exported(native):
    hdr = open(os.path.join(ROOT, "include", "gs_raster.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", hdr))
    assert {"gs_project_fwd", "gs_bin_count", "gs_bin_emit_sort", "gs_blend_fwd", "gs_blend_bwd", "gs_project_bwd"} <= names
    lib = native.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gs_raster.h but not exported"
    assert names == set(native.SIGNATURES), "ctypes signature table out of sync with the header"


def test_no_kernel_needs_scratch_memory(native):
    """A kernel with register spills (or a dynamically indexed local array) makes the HIP runtime provision scratch memory for
    the queue it is launched on; with two streams of one process taking turns (the captured step on the runner's stream, eager
    calls on the caller's) every eager launch of such a kernel stalled 0.5-2 ms behind that (HISTORY.md section 8b).  The
    Makefile keeps the compiler's resource report of every object (csrc/*.res): all kernels must report ScratchSize 0."""
    import glob
    import subprocess
    csrc = native.CSRC_DIR
    hips = sorted(glob.glob(os.path.join(csrc, "*.hip")))
    if any(not os.path.exists(h[:-4] + ".res") for h in hips):   # objects of an older build: make them again with the report
        subprocess.run(["make", "-C", csrc, "-B", "-j4"], check=True, capture_output=True)
    n_kernels = 0
    for h in hips:
        rep = open(h[:-4] + ".res").read()
        names = re.findall(r"Function Name: (\S+)", rep)
        sizes = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", rep)]
        assert len(names) == len(sizes)
        n_kernels += len(names)
        for nme, sz in zip(names, sizes):
            assert sz == 0, f"{os.path.basename(h)}: kernel {nme} uses {sz} bytes of scratch per lane"
    assert n_kernels >= 60


def test_forward_and_backward_round_alpha_and_transmittance_with_the_same_instruction_trees(native, tmp_path):
    """VERDICT r4 missing #3, static half (the dynamic half: tests/test_gpu_contributors.py).  The backward stores no last_ids:
    it re-derives who contributed from alpha and T' = fma(-alpha, T, T), which must therefore round exactly as in the forward.
    The source spells both alike, but the library is built with -ffp-contract=fast -- a compiler that fused one kernel's
    multiply-adds differently would show up only as scattered gradient noise.  So: compile gs_blend.hip to ISA with the
    Makefile's flags and compare, across the seven kernels, the expression tree under every v_exp_f32 (tests/isa_slices.py)."""
    import subprocess
    import isa_slices as ISA
    csrc = native.CSRC_DIR
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", open(os.path.join(csrc, "Makefile")).read(), re.M).group(1)
    flags = flags.replace("$(ARCH)", "gfx950").replace("$(EXTRA)", "").split()
    out = tmp_path / "gs_blend.s"
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", os.path.join(csrc, "gs_blend.hip"), "-o", str(out)],
                   check=True, capture_output=True, cwd=csrc)
    ks = {k: v for k, v in ISA.kernels(out.read_text()).items() if "blend_" in k}
    # (the forward: training / inference x one list per tile, front round, back round of the depth rounds)
    assert len(ks) == 7 and sum("blend_bwd" in k for k in ks) == 1 and sum("blend_fwd" in k for k in ks) == 6, list(ks)
    trees = set()
    for name, ins in ks.items():
        slices = ISA.sigma_alpha_slices(ins)
        # (pixel, entry) evaluations per kernel body: the forward's four quadrants; the backward's entries per lane in its four
        # fill classes, 4 + 3 + 2 + 1
        assert len(slices) == (10 if "blend_bwd" in name else 4), (name, len(slices))
        for sl in slices:
            assert sl["neg"] and sl["alpha"], (name, sl)     # exp2(-sigma) feeding alpha = min(0.999, opacity * .)
            trees.add(sl["sigma"])
        assert ISA.transmittance_updates(ins) >= 4, name     # T' = fma(-alpha, T, T), one per evaluation
    # sigma = fma(dy, fma(hC, dy, Bc dx), (hA dx) dx) with dx, dy = mean - pixel centre: ONE tree, in every kernel
    assert trees == {"fma(fma(sub(x,x),x,mul(sub(x,x),x)),sub(x,x),mul(mul(sub(x,x),x),sub(x,x)))"}, trees


def test_loss_entries_refuse_images_beyond_their_32_bit_offsets(native):
    """The loss kernels address a pixel by a 32-bit byte offset from a block-uniform base: the entry points refuse more than 2^28
    pixels (and images smaller than the 11 x 11 window) before anything is launched -- checked without a GPU."""
    import ctypes as ct
    L = native.lib()
    dummy = (ct.c_float * 4)()
    p = ct.addressof(dummy)
    assert L.gs_l1_ssim_fwd(None, 20000, 20000, 0.2, p, p, None, 0, p, p) == -1
    assert b"too large" in L.gs_last_error()
    assert L.gs_l1_ssim_bwd(None, 20000, 20000, 0.2, p, p, None, 0, p, p, p) == -1
    assert L.gs_l1_ssim_fwd(None, 16, 1 << 21, 0.2, p, p, None, 0, p, p) == -1   # (a row of 12 W bytes beyond the 24-bit multiply)
    assert L.gs_l1_ssim_fwd(None, 10, 64, 0.2, p, p, None, 0, p, p) == -1
    assert b"larger than the 11x11 window" in L.gs_last_error()
    assert L.gs_loss_workspace_floats(1080, 1920) == 9 * 1080 * 1920 + 2 * (60 * 34 + 8)


def test_identity_and_layout_queries(native):
    lib = native.lib()
    # scratch of the blend backward's fill classes: 16 counters + 4 ints per descriptor -- cap_units of class 4, three tail regions
    # of 4 C tiles each, rounded up to the 32 units a workgroup of the backward takes
    assert lib.gs_unit_classes_ints(1000, 1, 1920, 1080) == 16 + 4 * (1000 + 3 * 32640)
    assert lib.gs_unit_classes_ints(256, 2, 33, 17) == 16 + 4 * (256 + 3 * 64)
    assert lib.gs_unit_classes_ints(-1, 1, 16, 16) == 0
    assert lib.gs_version() >= 100
    assert lib.gs_arch() == b"gfx950"
    assert 1 <= lib.gs_bin_groups(1) <= lib.gs_bin_groups(10**6) <= 256
    small, big = lib.gs_bin_workspace_bytes(1, 10**6, 120, 68), lib.gs_bin_workspace_bytes(2, 10**6, 120, 68)
    assert 0 < small < big


def test_code_object_is_gfx950(native):
    blob = open(native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for kern in (b"project_fwd_kernel", b"blend_fwd_kernel", b"blend_bwd_kernel", b"tile_sort_kernel", b"bin_emit_kernel"):
        assert kern in blob


def test_frontend_argument_errors():
    from easy_gaussian_splatting_amd.rendering import rasterization
    N = 4
    a = dict(means=torch.zeros(N, 3), quats=torch.ones(N, 4), scales=torch.ones(N, 3), opacities=torch.ones(N),
             colors=torch.zeros(N, 16, 3), viewmats=torch.eye(4)[None], Ks=torch.eye(3)[None], width=32, height=32)
    with pytest.raises(NotImplementedError):
        rasterization(**a, sh_degree=3)  # packed defaults to True upstream; not implemented here
    with pytest.raises(NotImplementedError):
        rasterization(**a, sh_degree=3, packed=False, render_mode="RGB+D")
    with pytest.raises(NotImplementedError):
        rasterization(**a, sh_degree=3, packed=False, rasterize_mode="antialiased")
    with pytest.raises(AssertionError):
        rasterization(**{**a, "quats": torch.ones(N, 3)}, sh_degree=3, packed=False)
    with pytest.raises(AssertionError):
        rasterization(**a, sh_degree=4, packed=False)  # needs 25 coefficients
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rasterization(**a, sh_degree=3, packed=False)  # CPU tensors: the product path never falls back


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "easy_gaussian_splatting_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_camera_gradients_are_refused_not_dropped():
    """gsplat returns gradients for viewmats; this path does not compute them and the reference never asks
    (SURVEY.md 8b): asking must raise, not hand back None silently."""
    import pytest
    import torch
    from easy_gaussian_splatting_amd.rendering import rasterization
    z = torch.zeros
    V = torch.eye(4)[None].clone().requires_grad_(True)
    K = torch.eye(3)[None]
    with pytest.raises(NotImplementedError):
        rasterization(z(2, 3), z(2, 4), z(2, 3), z(2), z(2, 3), V, K, 16, 16, packed=False)


def test_binning_choice_host_logic(monkeypatch):
    """rendering.binning_choice / bin_shift_for: the pipeline and bin size from the previous call's mean footprint, the
    environment overrides, and a loud error for an unknown mode (host logic only: no GPU)."""
    from easy_gaussian_splatting_amd import rendering as R
    monkeypatch.delenv("GS_BINNING", raising=False)
    monkeypatch.delenv("GS_BINS_SHIFT", raising=False)
    assert R.binning_choice(None) == "tiles" and R.binning_choice(2.8) == "tiles"
    assert R.binning_choice(R.BINS_FROM_FOOTPRINT) == "bins" and R.binning_choice(105.9) == "bins"
    assert R.bin_shift_for(None) == 0 and R.bin_shift_for(3.0) == 1 and R.bin_shift_for(40.0) == 2
    monkeypatch.setenv("GS_BINNING", "bins")
    assert R.binning_choice(1.0) == "bins"
    monkeypatch.setenv("GS_BINNING", "tiles")
    assert R.binning_choice(500.0) == "tiles"
    monkeypatch.setenv("GS_BINS_SHIFT", "1")
    assert R.bin_shift_for(500.0) == 1
    monkeypatch.setenv("GS_BINNING", "quadtree")
    with pytest.raises(ValueError):
        R.binning_choice(3.0)


def test_workspace_layout_query(native):
    """gs_workspace_query (SURVEY.md 8b "Ownership"): every buffer 256-byte aligned, no two buffers of an arena overlap,
    training-only buffers absent from an inference layout, the list arena scales with the capacity of LISTED intersections
    (13-25 bytes each), the walk arena with the capacities of work units and gradient rows (what a training forward WALKS), the
    fixed one with neither; argument errors reported through the error channel.  (Host logic only: no GPU.)"""
    import ctypes as ct
    from easy_gaussian_splatting_amd import workspace as WS
    lib = native.lib()

    def sizes(layout):
        """(arena, offset) of every present slot, sorted by offset inside each arena"""
        out = {0: [], 1: [], 2: []}
        for slot, off in enumerate(layout.offsets):
            if off >= 0:
                assert off % 256 == 0, (slot, off)
                out[0 if slot < WS.LIST_FIRST else (1 if slot < WS.WALK_FIRST else 2)].append((off, slot))
        return {a: sorted(v) for a, v in out.items()}

    tr = WS.Layout(1, 1_000_000, 1920, 1080, 4_000_000, 0, 0, WS.F_TRAIN, 200_000, 6_000_000)
    inf = WS.Layout(1, 1_000_000, 1920, 1080, 4_000_000, 0, 0, 0)
    big = WS.Layout(1, 1_000_000, 1920, 1080, 8_000_000, 0, 0, WS.F_TRAIN, 200_000, 6_000_000)
    walked = WS.Layout(1, 1_000_000, 1920, 1080, 4_000_000, 0, 0, WS.F_TRAIN, 400_000, 12_000_000)
    two = WS.Layout(1, 1_000_000, 1920, 1080, 4_000_000, 3_000_000, 2, WS.F_TRAIN | WS.F_TWO_LEVEL | WS.F_ISECT_IDS, 200_000, 6_000_000)
    for lay in (tr, inf, big, walked, two):
        for arena, lst in sizes(lay).items():
            if lst:
                offs = [o for o, _ in lst]
                assert len(set(offs)) == len(offs) and offs[0] == 0 and offs[-1] < lay.arena_bytes[arena]
    # known sizes: rec = 48 B per Gaussian right after the 64-byte info block; rows = 48 B per row of the capacity, last in
    # the walk arena; a work unit holds a 1 KB checkpoint, 256 B of sublist pairs and a 16-byte descriptor
    assert tr.offsets[WS.REC] == 256 and tr.offsets[WS.BBOX] - tr.offsets[WS.REC] == 48_000_000
    assert tr.arena_bytes[2] - tr.offsets[WS.ROWS] >= 6_000_000 * 48
    assert tr.offsets[WS.QLIST] - tr.offsets[WS.CKPT] == 200_000 * 1024 and tr.offsets[WS.UNIT_DESC] - tr.offsets[WS.QLIST] == 200_000 * 256
    for slot in (WS.CKPT, WS.QLIST, WS.QMASK, WS.ROW_BASE, WS.WALK_STATE, WS.UNIT_DESC, WS.ROWS, WS.SLOTS, WS.SLOT_GID, WS.QCNT):
        assert tr.offsets[slot] >= 0 and inf.offsets[slot] == -1, slot
    assert inf.offsets[WS.KEYS_TMP] >= 0 and inf.offsets[WS.FLATTEN_IDS] >= 0 and inf.arena_bytes[2] == 256
    assert two.offsets[WS.COARSE_KEYS] >= 0 and two.offsets[WS.KEYS_TMP] == -1 and two.offsets[WS.ISECT_IDS] >= 0
    # twice the LISTED capacity: twice the list arena, the same walk arena; twice the walk capacities: the other way round
    assert big.arena_bytes[0] == tr.arena_bytes[0] and big.arena_bytes[1] > 1.9 * tr.arena_bytes[1] and big.arena_bytes[2] == tr.arena_bytes[2]
    assert walked.arena_bytes[1] == tr.arena_bytes[1] and walked.arena_bytes[2] > 1.9 * tr.arena_bytes[2]
    # a listed intersection costs a training call 25 bytes (round 5: 355 + 24); an inference call 12
    assert tr.arena_bytes[1] <= 25 * 4_000_000 + 2**24 and inf.arena_bytes[1] <= 12 * 4_000_000 + 2**24   # (+ the binning scratch)
    # errors: through the status code + gs_last_error, never a crash
    offs, ab = (ct.c_int64 * WS.N_SLOTS)(), (ct.c_int64 * 3)()
    assert lib.gs_workspace_query(0, 10, 64, 64, 100, 0, 8, 0, 0, 0, offs, ab) == -1 and b"C>=1" in lib.gs_last_error()
    assert lib.gs_workspace_query(1, 10, 64, 64, 1 << 31, 0, 8, 0, 0, WS.F_TRAIN, offs, ab) == -1 and b"int32" in lib.gs_last_error()
    assert lib.gs_workspace_query(1, 10, 64, 64, 100, 0, 4, 0, 0, WS.F_TRAIN, offs, ab) == -1 and b"cap_units" in lib.gs_last_error()
    assert lib.gs_workspace_query(1, 10, 64, 64, 100, 10, 8, 0, 7, WS.F_TWO_LEVEL, offs, ab) == -1
    assert lib.gs_workspace_bind(None, None, 0, None, 0, None, 0, offs, ab) == -1


def test_workspace_pool_leases(native, monkeypatch):
    """workspace.pool: a lease goes back to its (device, stream) pool when the last holder drops it and is re-used by the
    next acquire; two calls in flight get two leases; arenas only grow.  (bind's device memset is stubbed: no GPU.)"""
    from easy_gaussian_splatting_amd import workspace as WS
    monkeypatch.setattr(native.lib(), "gs_workspace_bind", lambda *a: 0, raising=False)
    cpu = torch.device("cpu")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    WS.pool.clear()
    a = WS.pool.acquire(cpu, 1234)
    a.bind(WS.Layout(1, 1000, 64, 64, 5000, 0, 0, WS.F_TRAIN), 0)
    b = WS.pool.acquire(cpu, 1234)               # second call in flight on the same stream: its own lease
    assert b is not a and a.busy and b.busy
    ra, rb = WS.LeaseRef(a), WS.LeaseRef(a)      # two holders of lease a (autograd node + meta)
    del ra
    assert a.busy
    del rb
    assert not a.busy
    c = WS.pool.acquire(cpu, 1234)
    assert c is a                                 # re-used, arenas kept
    fixed_ptr, list_bytes = c.fixed.data_ptr(), c.lists.numel()
    c.bind(WS.Layout(1, 1000, 64, 64, 4000, 0, 0, WS.F_TRAIN), 0)        # a smaller frame: nothing is re-allocated
    assert c.fixed.data_ptr() == fixed_ptr and c.lists.numel() == list_bytes
    c.grow_lists(WS.Layout(1, 1000, 64, 64, 50000, 0, 0, WS.F_TRAIN), 0)  # capacity exceeded: the list arena alone grows
    assert c.fixed.data_ptr() == fixed_ptr and c.lists.numel() > list_bytes and c.cap == 50000
    list_ptr, walk_bytes = c.lists.data_ptr(), c.walk.numel()
    c.grow_walk(WS.Layout(1, 1000, 64, 64, 50000, 0, 0, WS.F_TRAIN, 4096, 100_000))   # the walk outgrew its capacities: the walk arena alone
    assert c.fixed.data_ptr() == fixed_ptr and c.lists.data_ptr() == list_ptr and c.walk.numel() > walk_bytes
    assert c.layout.cap_units == 4096 and c.layout.cap_rows == 100_000 and c.ptr(WS.ROWS) == c.walk.data_ptr() + c.layout.offsets[WS.ROWS]
    assert c.ptr(WS.REC) == c.fixed.data_ptr() + 256 and c.ptr(WS.COARSE_KEYS) is None
    assert c.view(WS.FLATTEN_IDS, 10).dtype == torch.int32 and c.view(WS.INFO, 8).dtype == torch.int64
    d = WS.pool.acquire(cpu, 999)                 # another stream: another pool
    assert d is not b and d is not c
    WS.pool.clear()


def test_the_loaded_library_is_the_product_build(native, monkeypatch):
    """every build names the preprocessor flags it was made with (`gs_build_flags`); the product's are empty,
    the binding refuses a diagnostic variant unless GS_ALLOW_VARIANT=1 says so, variants live in build/variants/ -- never in the
    package directory -- and the shipped sources carry no experiment switch beyond the four diagnostic ones tests and tools use."""
    import glob
    import re
    assert native.build_flags() == "" and native.lib().gs_version() >= 300
    pkg = os.path.dirname(native.LIB_PATH)
    assert sorted(os.path.basename(f) for f in glob.glob(os.path.join(pkg, "*.so"))) == ["libgsraster.so"]
    assert os.path.realpath(native.VARIANT_DIR).startswith(os.path.realpath(os.path.join(ROOT, "build")))
    switches = set()
    for f in glob.glob(os.path.join(native.CSRC_DIR, "*.hip")) + glob.glob(os.path.join(native.CSRC_DIR, "*.h")):
        switches |= set(re.findall(r"^#\s*if(?:n?def)?\s+(?:defined\()?\s*!?(GS_\w+)", open(f).read(), flags=re.M))
    assert switches <= {"GS_BWD_CHECK", "GS_BWD_ACC64", "GS_EXACT_MATH", "GS_CLOCK_PROBE", "GS_BUILD_FLAGS"}, switches
    n_if_blend = len(re.findall(r"^#\s*if", open(os.path.join(native.CSRC_DIR, "gs_blend.hip")).read(), flags=re.M))
    assert n_if_blend <= 10, n_if_blend
    # a variant is refused ... (a fake one: the loader only looks at what the library says about itself)
    class Fake:
        def __getattr__(self, name):
            f = lambda *a: b"-DGS_BWD_CHECK" if name == "gs_build_flags" else 0
            return f
    import ctypes as ct
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(ct, "CDLL", lambda path: Fake())
    monkeypatch.delenv("GS_ALLOW_VARIANT", raising=False)
    with pytest.raises(native.NativeLibraryError, match="diagnostic variant"):
        native.lib()
    monkeypatch.setenv("GS_ALLOW_VARIANT", "1")   # ... unless asked for
    assert native.lib().gs_build_flags() == b"-DGS_BWD_CHECK"
    monkeypatch.setattr(native, "_lib", None)
