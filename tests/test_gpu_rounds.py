"""Depth rounds (include/gs_raster.h "Depth rounds", gs_rounds.hip): the list stages and the blend forward run the frame in two
rounds -- the front slab by depth, then the rest only into tiles the front slab has not finished.  Every tile walks the same
entries in the same order as over one list, so the contract is the strongest there is: images, loss, quadrant sublists and
the SET of gradient rows bit for bit those of the one-round pipeline (which tests/test_gpu_parity.py holds against the oracle);
the per-Gaussian gradient sums equal to the rounding of a different summation tree (a wave's rows are read as two ranges)."""
import numpy as np
import pytest
import torch

from easy_gaussian_splatting_amd import _native as nat
from easy_gaussian_splatting_amd.loss import LossComputer
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers
from easy_gaussian_splatting_amd.synthetic import config_heavy
from easy_gaussian_splatting_amd.train_graph import TrainStepGraph
from scenes import make_scene

pytestmark = pytest.mark.gpu
LRS = (1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)


def _scene(kind: str):
    if kind == "sparse":   # few tiles per Gaussian, most tiles still live behind the front slab: the back round does the work
        return make_scene(20000, 320, 208, sh_degree=3, n_views=2, seed=3, scale_range=(0.01, 0.08), dist=4.0)
    # heavy-tailed sizes (synthetic.config_heavy at a small image): the front slab saturates most of the image
    return config_heavy(seed=5, n=40000, n_views=2, width=480, height=272, median=0.05)


def _setup(kind: str):
    dev = torch.device("cuda:0")
    sc = _scene(kind)
    W, H = int(sc["width"]), int(sc["height"])
    T = torch.from_numpy
    op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
    shs = T(sc["shs"])

    def make():
        m = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]),
                          sh_0=shs[:, :1].contiguous(), sh_rest=shs[:, 1:].contiguous(),
                          logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3,
                          white_background=bool(np.all(sc["backgrounds"] == 1.0)), means_lr_schedule_max_steps=40).to(dev)
        return m, build_optimizers(m, *LRS, fused="hip")

    n_views = sc["viewmats"].shape[0]
    datas = [{"w2c": T(sc["viewmats"][v]).to(dev), "K": T(sc["Ks"][v]).to(dev), "width": W, "height": H} for v in range(n_views)]
    g = torch.Generator().manual_seed(11)
    gts = [torch.rand((H, W, 3), generator=g).to(dev) for _ in range(n_views)]
    return dev, make, datas, gts


def _close(a, b, rtol, what):
    a, b = a.double(), b.double()
    scale = max(float(a.abs().max()), 1e-30)
    err = float((a - b).abs().max()) / scale
    assert err <= rtol, (what, err)


@pytest.mark.parametrize("binning", ["tiles", "bins"])
@pytest.mark.parametrize("kind,fraction", [("sparse", 0.125), ("sparse", 0.5), ("heavy", 0.125), ("heavy", 0.03)])
@pytest.mark.parametrize("use_graph", [False, True])
def test_two_rounds_equal_one_round(kind, fraction, binning, use_graph, monkeypatch):
    monkeypatch.setenv("GS_BINNING", binning)
    dev, make, datas, gts = _setup(kind)
    lc = LossComputer(0.2, clamp_input=True)
    for v in (0, 1):   # (fresh models per view: a step later the two parameter sets differ by the rounding of their gradient sums)
        (ma, oa), (mb, ob) = make(), make()
        ra = TrainStepGraph(ma, oa, lc, datas[v], gts[v], use_graph=use_graph, fuse_adam=False, rounds="off")
        rb = TrainStepGraph(mb, ob, lc, datas[v], gts[v], use_graph=use_graph, fuse_adam=False, rounds="on", round_fraction=fraction)
        assert rb.report()["rounds"] and not ra.report()["rounds"] and rb.report()["binning"] == binning
        outa = ra.step(datas[v], gts[v]); ra.finish()
        outb = rb.step(datas[v], gts[v]); rb.finish()
        torch.cuda.synchronize()
        assert torch.equal(outa["render_img"], outb["render_img"]), "image"
        assert torch.equal(ra.buf["render_alphas"], rb.buf["render_alphas"]), "alphas"
        assert torch.equal(outa["loss3"], outb["loss3"]), "loss"
        assert torch.equal(ra.buf["qcnt"], rb.buf["qcnt"]), "quadrant sublist lengths"
        wa, wb = ra.buf["walk_state"][:8].tolist(), rb.buf["walk_state"][:8].tolist()
        assert wa[nat_walk("UNITS")] == wb[nat_walk("UNITS")] and wa[nat_walk("ROWS")] == wb[nat_walk("ROWS")], (wa, wb)
        blk = rb.buf["rounds"].tolist()
        listed_a, listed_b = int(ra.buf["info"][0]), int(rb.buf["info"][0])
        assert 0 < blk[nat.GS_ROUND_BASE] <= listed_b <= listed_a, (blk, listed_a, listed_b)
        # the rows themselves are the same set: their sums over everything agree to rounding
        na, nb = wa[nat_walk("ROWS")], wb[nat_walk("ROWS")]
        _close(ra.buf["rows"][:na].double().sum(0), rb.buf["rows"][:nb].double().sum(0), 1e-9, "row totals")
        # absgrad side channel and every parameter gradient: equal to the rounding of another summation order
        _close(outa["absgrad"], outb["absgrad"], 2e-5, "absgrad")
        for k, ga in ra.grads.items():
            if ga is not None:
                _close(ga, rb.grads[k], 2e-5, k)
        if kind == "sparse":
            assert blk[nat.GS_ROUND_LIVE] > 0 and listed_b > blk[nat.GS_ROUND_BASE], "the back round was expected to have work here"
        elif fraction >= 0.1:   # (a 3 % slab finishes no tile of this scene: every footprint survives the windowing)
            assert listed_b < listed_a, "the heavy scene was expected to list less in two rounds"


def nat_walk(name: str) -> int:
    return {"UNITS": 0, "STORAGE": 1, "ROWS": 3, "FLAGS": 4}[name]


@pytest.mark.parametrize("kind", ["sparse", "heavy"])
def test_rounds_trajectory_follows_one_round(kind):
    """Fused Adam, captured, six steps over two views: the two-round runner stays on the one-round trajectory (losses to 1e-5;
    not bitwise: Adam sees gradient sums of another summation order)."""
    dev, make, datas, gts = _setup(kind)
    (ma, oa), (mb, ob) = make(), make()
    lc = LossComputer(0.2, clamp_input=True)
    ra = TrainStepGraph(ma, oa, lc, datas[0], gts[0], rounds="off")
    rb = TrainStepGraph(mb, ob, lc, datas[0], gts[0], rounds="on")
    for it in range(6):
        v = it % 2
        ma.update_learning_rate(it); mb.update_learning_rate(it)
        la = ra.step(datas[v], gts[v])["loss3"].clone(); ra.finish()
        lb = rb.step(datas[v], gts[v])["loss3"].clone(); rb.finish()
        assert torch.allclose(la, lb, rtol=1e-5, atol=1e-7), (it, la, lb)
    for k in ma.param_names:
        _close(getattr(ma, k).detach(), getattr(mb, k).detach(), 1e-4, k)
    assert rb.report()["overflows"] == 0 and rb.report()["steps"] == 6


def test_rounds_survive_list_and_walk_overflows():
    """Capacities learnt on a far view, then close-ups: the overflowing two-round step is a device-side no-op, recovered like
    any other; the end state is the one-round runner's (to rounding)."""
    dev, make, datas, gts = _setup("sparse")
    far = dict(datas[0])
    w2c = far["w2c"].clone()
    w2c[2, 3] += 14.0
    far["w2c"] = w2c
    (ma, oa), (mb, ob) = make(), make()
    lc = LossComputer(0.2, clamp_input=True)
    ra = TrainStepGraph(ma, oa, lc, far, gts[0], margin=1.02, check_every=4, rounds="off")
    rb = TrainStepGraph(mb, ob, lc, far, gts[0], margin=1.02, check_every=4, rounds="on")
    seq = [far, datas[1], datas[0], far, datas[0], datas[1]]
    for it, d in enumerate(seq):
        ra.step(d, gts[it % 2]); rb.step(d, gts[it % 2])
    ra.finish(); rb.finish()
    assert rb.report()["overflows"] >= 1 and rb.report()["steps"] == len(seq)
    for k in ma.param_names:
        _close(getattr(ma, k).detach(), getattr(mb, k).detach(), 1e-4, k)


def test_auto_mode_turns_rounds_on_only_where_lists_dwarf_the_walk():
    dev, make, datas, gts = _setup("sparse")
    m, o = make()
    r = TrainStepGraph(m, o, LossComputer(0.2, clamp_input=True), datas[0], gts[0])
    assert not r.report()["rounds"]


# ---- the two passes that exist only with rounds, against numpy ------------------------------------------------------------
def _window_ref(fp, live, tw):
    x0, x1, y0, y1 = fp[0] & 0xffff, fp[0] >> 16, fp[1] & 0xffff, fp[1] >> 16
    w, h = x1 - x0, y1 - y0
    if fp[3] == 0:
        return (0, 0, 0, 0)
    if w * h <= 32:
        m = 0
        for i in range(w * h):
            if (fp[2] >> i) & 1 and live[y0 + i // w, x0 + i % w]:
                m |= 1 << i
        return (fp[0], fp[1], m, bin(m).count("1")) if m else (0, 0, 0, 0)
    sub = live[y0:y1, x0:x1]
    if not sub.any():
        return (0, 0, 0, 0)
    ys, xs = np.nonzero(sub)
    nx0, nx1, ny0, ny1 = x0 + xs.min(), x0 + xs.max() + 1, y0 + ys.min(), y0 + ys.max() + 1
    w, h = nx1 - nx0, ny1 - ny0
    if w * h <= 32:
        m = 0
        for i in range(w * h):
            if live[ny0 + i // w, nx0 + i % w]:
                m |= 1 << i
        return (nx0 | (nx1 << 16), ny0 | (ny1 << 16), m, bin(m).count("1"))
    return (nx0 | (nx1 << 16), ny0 | (ny1 << 16), 0xffffffff, w * h)


@pytest.mark.parametrize("tw,th", [(20, 13), (120, 68), (240, 135)])
def test_split_and_footprints_against_numpy(tw, th):
    dev = torch.device("cuda:0")
    L = nat.lib()
    rng = np.random.default_rng(tw)
    N = 6000
    x0 = rng.integers(0, tw, N); y0 = rng.integers(0, th, N)
    big = rng.random(N) < 0.3
    w = np.where(big, rng.integers(1, tw + 1, N), rng.integers(1, 7, N)); h = np.where(big, rng.integers(1, th + 1, N), rng.integers(1, 6, N))
    x1 = np.minimum(x0 + w, tw); y1 = np.minimum(y0 + h, th)
    rect = (x1 - x0) * (y1 - y0)
    mask = np.where(rect <= 32, rng.integers(0, 2 ** 32, N, dtype=np.uint64) & ((1 << np.minimum(rect, 32).astype(np.uint64)) - 1).astype(np.uint64), 0xffffffff)
    mask = np.where(rect == 32, rng.integers(0, 2 ** 32, N, dtype=np.uint64), mask).astype(np.uint64)
    cnt = np.where(rect <= 32, [bin(int(m)).count("1") for m in mask], rect)
    invisible = rng.random(N) < 0.1
    cnt = np.where(invisible, 0, cnt)
    fp = np.stack([x0 | (x1 << 16), y0 | (y1 << 16), mask, cnt], 1).astype(np.uint32)
    fp[cnt == 0] = 0
    depths = np.exp(rng.normal(1.0, 0.8, N)).astype(np.float32)
    live = rng.random((th, tw)) < 0.15
    live[: th // 3] = False
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)   # noqa: E731
    bbox, dep, tpg = t(fp.view(np.int32), torch.int32), t(depths, torch.float32), t(fp[:, 3].astype(np.int32), torch.int32)
    blk = torch.zeros(nat.GS_ROUND_WORDS, dtype=torch.int64, device=dev)
    hist = torch.zeros(4096, dtype=torch.int32, device=dev)
    tile_live = t(live.reshape(-1).astype(np.uint8), torch.uint8)
    state = torch.zeros((tw * th, 4, 64, 4), device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    frac = 0.3
    nat.check(L.gs_round_split(st, N, dep.data_ptr(), tpg.data_ptr(), frac, hist.data_ptr(), blk.data_ptr()), "gs_round_split")
    split = int(blk[nat.GS_ROUND_SPLIT])
    assert int(hist.abs().sum()) == 0
    # numpy: first 1/16-octave bin edge behind which >= frac of the weight lies
    bits = depths.view(np.uint32) >> 19
    wts = np.bincount(bits, weights=fp[:, 3].astype(np.float64), minlength=4096)
    cum = np.cumsum(wts)
    b = int(np.nonzero((cum >= np.ceil(frac * cum[-1])) & (wts > 0))[0][0])
    assert split == (b + 1) << 19
    front = (depths.view(np.uint32) < split) | (fp[:, 3] == 0)
    out = torch.full((N, 4), -1, dtype=torch.int32, device=dev)
    tpg_r = torch.full((N,), -7, dtype=torch.int32, device=dev)
    try:
        blk[nat.GS_ROUND_LIVE] = int(live.sum())
        for phase in (1, 2):
            nat.check(L.gs_rounds_set(blk.data_ptr(), tile_live.data_ptr(), state.data_ptr(), None, phase), "gs_rounds_set")
            nat.check(L.gs_round_footprints(st, N, tw, th, bbox.data_ptr(), dep.data_ptr(), out.data_ptr(), tpg_r.data_ptr()), "gs_round_footprints")
            got = out.cpu().numpy().view(np.uint32)
            if phase == 1:
                want = np.where(front[:, None], fp, 0)
                assert np.array_equal(got, want)
                assert np.array_equal(tpg_r.cpu().numpy(), want[:, 3].astype(np.int32))
                assert int(blk[nat.GS_ROUND_FRONT_N]) == int(((fp[:, 3] > 0) & front).sum())
            else:
                want = np.array([(0, 0, 0, 0) if front[i] else _window_ref([int(v) for v in fp[i]], live, tw) for i in range(N)], dtype=np.uint32)
                assert np.array_equal(got, want)
                want_tpg = np.where(front, fp[:, 3], want[:, 3]).astype(np.int32)
                assert np.array_equal(tpg_r.cpu().numpy(), want_tpg)
    finally:
        L.gs_rounds_set(None, None, None, None, 0)


# ---- the eager seam: inference calls in two rounds ---------------------------------------------------------------------------
@pytest.mark.parametrize("culling", ["gsplat", "tight", "gsplat_eager"])
@pytest.mark.parametrize("kind", ["sparse", "heavy"])
def test_inference_in_two_rounds_is_the_one_round_image(kind, culling):
    from easy_gaussian_splatting_amd import rendering
    dev = torch.device("cuda:0")
    sc = _scene(kind)
    W, H = int(sc["width"]), int(sc["height"])
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    args = (t["means"], t["quats"], t["scales"], t["opacities"], t["shs"], t["viewmats"][:1], t["Ks"][:1], W, H)
    kw = dict(sh_degree=3, packed=False, backgrounds=t["backgrounds"][:1], _tile_culling=culling)
    with torch.no_grad():
        rendering.reset_hints()
        n0 = rendering.stats["round_calls"]
        for _ in range(2):   # (second call: capacities learnt from the first)
            img_a, al_a, meta_a = rendering.rasterization(*args, _rounds="off", **kw)
        assert rendering.stats["round_calls"] == n0
        rendering.reset_hints()
        for _ in range(3):
            img_b, al_b, meta_b = rendering.rasterization(*args, _rounds="on", **kw)
        assert rendering.stats["round_calls"] == n0 + 3
        torch.cuda.synchronize()
        assert torch.equal(img_a, img_b) and torch.equal(al_a, al_b)
        for k in ("radii", "means2d", "depths", "conics", "tiles_per_gauss", "isect_offsets", "flatten_ids", "isect_ids"):
            assert torch.equal(meta_a[k], meta_b[k]), k
        # a call that needs gradients takes one round whatever is asked for
        ins = [t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "shs")]
        with torch.enable_grad():
            img_c, _, _ = rendering.rasterization(*ins, *args[5:], _rounds="on", **kw)
            img_c.sum().backward()
        assert rendering.stats["round_calls"] == n0 + 3 and torch.equal(img_c.detach(), img_a)
    rendering.reset_hints()


def test_inference_rounds_turn_themselves_on_for_long_lists():
    """`_rounds="auto"` (the default): the first call of a shape lists in one round; from 4 M listed intersections through the
    two-level binning upwards the following calls take two -- and keep doing so (their own short totals do not undo it)."""
    from easy_gaussian_splatting_amd import rendering
    dev = torch.device("cuda:0")
    sc = config_heavy(seed=5, n=300_000, n_views=1)
    W, H = int(sc["width"]), int(sc["height"])
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    args = (t["means"], t["quats"], t["scales"], t["opacities"], t["shs"], t["viewmats"], t["Ks"], W, H)
    kw = dict(sh_degree=3, packed=False, backgrounds=t["backgrounds"])
    with torch.no_grad():
        rendering.reset_hints()
        ref, _, meta = rendering.rasterization(*args, _rounds="off", **kw)
        listed = int(meta["flatten_ids"].numel())
        assert listed >= rendering.ROUNDS_MIN_LISTED, listed
        rendering.reset_hints()
        n0 = rendering.stats["round_calls"]
        outs = [rendering.rasterization(*args, **kw)[0] for _ in range(5)]
        torch.cuda.synchronize()
        assert rendering.stats["round_calls"] - n0 >= 3 and rendering.last_binning(dev) == "bins"
        for o in outs:
            assert torch.equal(o, ref)
    rendering.reset_hints()


# ---- corners ---------------------------------------------------------------------------------------------------------------
def _model_and_views(sc, dev, n_views):
    T = torch.from_numpy
    op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
    shs = T(sc["shs"])
    W, H = int(sc["width"]), int(sc["height"])

    def make():
        m = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]),
                          sh_0=shs[:, :1].contiguous(), sh_rest=shs[:, 1:].contiguous(),
                          logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=3, white_background=False).to(dev)
        return m, build_optimizers(m, *LRS, fused="hip")

    datas = [{"w2c": T(sc["viewmats"][v]).to(dev), "K": T(sc["Ks"][v]).to(dev), "width": W, "height": H} for v in range(n_views)]
    return make, datas


@pytest.mark.parametrize("W,H", [(2100, 300), (333, 517)])
def test_rounds_on_ragged_and_wide_tile_grids(W, H):
    """Tile grids that are not a multiple of anything: 132 x 19 tiles (three 64-bit words per row of the live-tile bitmap) and
    21 x 33 with ragged right / bottom tiles (pixels outside the image are finished from the start)."""
    dev = torch.device("cuda:0")
    sc = config_heavy(seed=9, n=30000, n_views=1, width=W, height=H, median=0.04)
    make, datas = _model_and_views(sc, dev, 1)
    gt = torch.rand((H, W, 3), generator=torch.Generator().manual_seed(2)).to(dev)
    lc = LossComputer(0.2, clamp_input=True)
    (ma, oa), (mb, ob) = make(), make()
    ra = TrainStepGraph(ma, oa, lc, datas[0], gt, fuse_adam=False, rounds="off")
    rb = TrainStepGraph(mb, ob, lc, datas[0], gt, fuse_adam=False, rounds="on", round_fraction=0.1)
    oa_, ob_ = ra.step(), rb.step()
    ra.finish(); rb.finish()
    torch.cuda.synchronize()
    assert torch.equal(oa_["render_img"], ob_["render_img"]) and torch.equal(oa_["loss3"], ob_["loss3"])
    assert torch.equal(ra.buf["qcnt"], rb.buf["qcnt"])
    for k, ga in ra.grads.items():
        if ga is not None:
            _close(ga, rb.grads[k], 2e-5, k)


def test_rounds_on_a_frame_that_lists_nothing():
    """Every Gaussian behind the camera: no weight in the depth histogram, no entry in either round; the step must still apply
    (Adam on zero gradients), exactly like the one-round step."""
    dev = torch.device("cuda:0")
    sc = _scene("sparse")
    sc["means"] = sc["means"].copy()
    sc["means"][:, 2] -= 100.0
    make, datas = _model_and_views(sc, dev, 1)
    W, H = int(sc["width"]), int(sc["height"])
    gt = torch.rand((H, W, 3), generator=torch.Generator().manual_seed(2)).to(dev)
    lc = LossComputer(0.2, clamp_input=True)
    (ma, oa), (mb, ob) = make(), make()
    ra = TrainStepGraph(ma, oa, lc, datas[0], gt, rounds="off")
    rb = TrainStepGraph(mb, ob, lc, datas[0], gt, rounds="on")
    for _ in range(2):
        la, lb = ra.step()["loss3"].clone(), rb.step()["loss3"].clone()
    ra.finish(); rb.finish()
    assert torch.equal(la, lb) and int(rb.buf["info"][0]) == 0 and rb.report()["steps"] == 2
    for k in ma.param_names:
        assert torch.equal(getattr(ma, k).detach(), getattr(mb, k).detach()), k


def test_rounds_through_refinement_and_an_opacity_reset():
    """The loop shape of the reference with rounds forced on: 60 steps over three views, densify_and_prune every 15, one opacity
    reset -- re-builds on projected capacities and behind a probe, with the round buffers re-made for the new N.  Against the
    one-round runner on the same schedule: the same number of Gaussians to 0.2 % (a rounding-level difference in a gradient norm
    can flip a densification decision), the loss to 1e-3, no overflow storm."""
    dev = torch.device("cuda:0")
    sc = config_heavy(seed=5, n=40000, n_views=3, width=480, height=272, median=0.05)
    make, datas = _model_and_views(sc, dev, 3)
    W, H = 480, 272
    with torch.no_grad():
        ref, _ = make()
        gts = [ref(d)["render_img"].clone() for d in datas]
    del ref

    def run(rounds):
        m, o = make()
        with torch.no_grad():
            m.means.add_(0.01 * torch.randn(m.means.shape, generator=torch.Generator().manual_seed(4)).to(dev))
        gen = torch.Generator(device=dev).manual_seed(9)
        r = TrainStepGraph(m, o, LossComputer(0.2, clamp_input=True), datas[0], gts[0], rounds=rounds)
        ns = []
        for it in range(1, 61):
            r.step(datas[it % 3], gts[it % 3])
            m.update_learning_rate(it)
            if it % 15 == 0 and it < 60:
                r.finish()
                if it == 30:
                    m.reset_opacities()
                else:
                    m.densify_and_prune(generator=gen)
                ns.append(m.nbr_gaussians)
        r.finish()
        return ns, float(r.loss_history(10)[:, 2].mean()), r.report()

    ns_a, la, rep_a = run("off")
    ns_b, lb, rep_b = run("on")
    assert rep_b["rounds"] and rep_b["steps"] == 60 and rep_b["overflows"] <= rep_a["overflows"] + 2, (rep_a, rep_b)
    for a, b in zip(ns_a, ns_b):
        assert abs(a - b) <= max(2, 0.002 * a), (ns_a, ns_b)
    assert abs(la - lb) <= 1e-3 * max(abs(la), 1e-6) + 1e-5, (la, lb)


# ---- "auto": the captured step holds the front round alone and is voided by a frame that needs the back round ------------------
def test_auto_captures_the_front_round_alone_and_recovers_when_a_frame_needs_more(monkeypatch):
    """`rounds="auto"` on a scene whose front slab finishes every tile of the probed view: the graph holds the front round only
    (no idle back round: `front_round_alone`).  Then a view arrives in which a quarter of the tiles see background only -- the
    front round leaves them live: the step must void itself on the device (GS_FLAG_BACK), be found at the next poll and be replayed after a
    re-build (which, on that view, turns rounds off) -- every step applied once, the end state the one-round runner's."""
    monkeypatch.setattr(TrainStepGraph, "ROUNDS_MIN_LISTED", 1000)   # (the test scene lists 0.38 M entries, 2.2 per gradient row)
    monkeypatch.setattr(TrainStepGraph, "ROUNDS_MIN_RATIO", 1.0)
    dev, make, datas, gts = _setup("heavy")
    far = dict(datas[0])
    w2c = far["w2c"].clone()
    w2c[0, 3] += 2.5   # (the scene slides to the right: a quarter of the tiles see background only)
    far["w2c"] = w2c
    (ma, oa), (mb, ob) = make(), make()
    lc = LossComputer(0.2, clamp_input=True)
    ra = TrainStepGraph(ma, oa, lc, datas[0], gts[0], check_every=4, rounds="off")
    rb = TrainStepGraph(mb, ob, lc, datas[0], gts[0], check_every=4, rounds="auto", round_fraction=0.5)
    rep = rb.report()
    assert rep["rounds"] and rep["front_round_alone"], rep
    for it in range(3):   # the speculation holds: same image and loss as one round
        la = ra.step(datas[0], gts[0])["loss3"].clone(); ra.finish()
        lb = rb.step(datas[0], gts[0])["loss3"].clone(); rb.finish()
        assert torch.allclose(la, lb, rtol=1e-5, atol=1e-7), (it, la, lb)
    assert rb.report()["overflows"] == 0
    seq = [datas[0], far, datas[0], far, datas[0], datas[0]]
    for d in seq:   # (the host keeps enqueueing behind the voided step)
        ra.step(d, gts[0]); rb.step(d, gts[0])
    ra.finish(); rb.finish()
    rep = rb.report()
    assert rep["steps"] == 3 + len(seq) and rep.get("back_round_needed", 0) >= 1 and not rep["front_round_alone"], rep
    for k in ma.param_names:
        _close(getattr(ma, k).detach(), getattr(mb, k).detach(), 2e-4, k)


# GS_FUZZ_CASES=N widens the sweep (as tests/test_gpu_parity.py::test_randomised_configurations, whose scenes these are)
@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("GS_FUZZ_CASES", "24"))))
def test_randomised_configurations_in_two_rounds(case, monkeypatch):
    """The randomised sweep of the parity suite (sizes, SH degree, stored K, background, splat scale, list mode, binning
    pipeline), first camera, forward: two depth rounds at a drawn split == one round, bit for bit -- image, alphas and the
    list arrays `meta` hands out."""
    from easy_gaussian_splatting_amd import rendering
    from test_gpu_parity import fuzz_case, to_dev
    monkeypatch.setenv("GS_BINNING", ("tiles", "bins", "bins")[case % 3])
    monkeypatch.setenv("GS_BINS_SHIFT", ("", "1", "2")[case % 3])
    sc, (deg, W, H, use_bg, split, culling) = fuzz_case(case, 0)
    t = to_dev(sc)
    frac = (0.05, 0.125, 0.3, 0.6)[case % 4]
    monkeypatch.setattr(rendering, "ROUND_FRACTION", frac)
    colors = (t["shs"][:, :1].contiguous(), t["shs"][:, 1:].contiguous()) if split else t["shs"]
    args = (t["means"], t["quats"], t["scales"], t["opacities"], colors, t["viewmats"][:1], t["Ks"][:1], W, H)
    kw = dict(sh_degree=deg, packed=False, backgrounds=t["backgrounds"][:1] if use_bg else None, _tile_culling=culling)
    with torch.no_grad():
        rendering.reset_hints()
        for _ in range(2):
            img_a, al_a, meta_a = rendering.rasterization(*args, _rounds="off", **kw)
        rendering.reset_hints()
        for _ in range(2):
            img_b, al_b, meta_b = rendering.rasterization(*args, _rounds="on", **kw)
        torch.cuda.synchronize()
        assert torch.equal(img_a, img_b) and torch.equal(al_a, al_b), (case, frac)
        for k in ("radii", "tiles_per_gauss", "isect_offsets", "flatten_ids"):
            assert torch.equal(meta_a[k], meta_b[k]), (case, k)
    rendering.reset_hints()


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("GS_FUZZ_CASES", "24"))))
def test_randomised_train_steps_in_two_rounds(case, monkeypatch):
    """The same sweep through the captured train step: one step in two rounds (split drawn per case) against one step in one round
    -- image, loss, sublist lengths, walk counters bitwise; every gradient to the rounding of another summation order."""
    from test_gpu_parity import fuzz_case
    monkeypatch.setenv("GS_BINNING", ("tiles", "bins", "bins")[case % 3])
    monkeypatch.setenv("GS_BINS_SHIFT", ("", "1", "2")[case % 3])
    sc, (deg, W, H, use_bg, split, culling) = fuzz_case(case, 0)
    dev = torch.device("cuda:0")
    T = torch.from_numpy
    op = np.clip(sc["opacities"], 1e-3, 1 - 1e-3)
    shs = T(sc["shs"])

    def make():
        m = GaussianModel(means=T(sc["means"]), log_scales=torch.log(T(sc["scales"])), quats=T(sc["quats"]),
                          sh_0=shs[:, :1].contiguous(), sh_rest=shs[:, 1:].contiguous(),
                          logit_opacities=T(np.log(op / (1 - op)).astype(np.float32)), sh_degree=deg,
                          white_background=bool(np.all(sc["backgrounds"] == 1.0))).to(dev)
        m.active_sh_degree = deg
        m.tile_culling = {"gsplat": "gsplat", "tight": "tight", "gsplat_eager": "gsplat_eager"}[culling]
        return m, build_optimizers(m, *LRS, fused="hip")

    data = {"w2c": T(sc["viewmats"][0]).to(dev), "K": T(sc["Ks"][0]).to(dev), "width": W, "height": H}
    gt = torch.rand((H, W, 3), generator=torch.Generator().manual_seed(case)).to(dev)
    lc = LossComputer(0.2, clamp_input=True)
    (ma, oa), (mb, ob) = make(), make()
    ra = TrainStepGraph(ma, oa, lc, data, gt, use_graph=bool(case & 1), fuse_adam=False, rounds="off")
    rb = TrainStepGraph(mb, ob, lc, data, gt, use_graph=bool(case & 1), fuse_adam=False, rounds="on", round_fraction=(0.05, 0.125, 0.3, 0.6)[case % 4])
    oa_, ob_ = ra.step(), rb.step()
    ra.finish(); rb.finish()
    torch.cuda.synchronize()
    assert torch.equal(oa_["render_img"], ob_["render_img"]) and torch.equal(oa_["loss3"], ob_["loss3"]), case
    assert torch.equal(ra.buf["qcnt"], rb.buf["qcnt"])
    wa, wb = ra.buf["walk_state"][:8].tolist(), rb.buf["walk_state"][:8].tolist()
    assert wa[0] == wb[0] and wa[3] == wb[3], (wa, wb)
    _close(oa_["absgrad"], ob_["absgrad"], 2e-5, "absgrad")
    for k, ga in ra.grads.items():
        if ga is not None and ga.numel():
            _close(ga, rb.grads[k], 2e-5, k)
