"""Row f-4: data formats either side of the hot path.  The image / mask helpers are pinned by vectors captured from
the reference's own scene/data_class.py (tests/golden/make_golden.py); the COLMAP / Blender readers and the checkpoint
layout are validated on synthetic files written by tools/make_synthetic_dataset.py against what the generator put in."""
import json
import os
import pickle
import random
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from make_synthetic_dataset import write_blender, write_colmap  # noqa: E402

from easy_gaussian_splatting_amd import checkpoint as ckpt  # noqa: E402
from easy_gaussian_splatting_amd import scene as S  # noqa: E402
from easy_gaussian_splatting_amd.model import GaussianModel, build_optimizers  # noqa: E402

GOLD = np.load(Path(__file__).parent / "golden" / "ref_data_class.npz")


def test_expand_mask_matches_reference_vectors():
    for i in range(3):
        for e in (0, 1, 2, 5):
            assert np.array_equal(S.expand_mask(GOLD[f"mask{i}"].copy(), e), GOLD[f"mask{i}_e{e}"]), (i, e)


def test_image_helpers_and_frame_match_reference_vectors(tmp_path):
    Image.fromarray(GOLD["rgba"], "RGBA").save(tmp_path / "a.png")
    assert np.array_equal(S.get_image_arr(tmp_path / "a.png", True), GOLD["rgba_white"])
    assert np.array_equal(S.get_image_arr(tmp_path / "a.png", False), GOLD["rgba_black"])
    Image.fromarray(GOLD["rgb"], "RGB").save(tmp_path / "b.png")
    Image.fromarray(GOLD["rgb_mask"], "L").save(tmp_path / "b_mask.png")
    fr = S.Frame(tmp_path / "b.png", tmp_path / "b_mask.png", 2, 64, 48, 70.0, 71.0, 31.5, 24.25, GOLD["frame_w2c"], False)
    d = fr.to_data()
    for k in ("K", "w2c", "image", "mask"):
        assert np.array_equal(d[k].numpy(), GOLD[f"frame_{k}"]), k
    assert [d["height"], d["width"]] == GOLD["frame_hw"].tolist()
    j = fr.to_json(3)
    assert np.allclose(j["position"], GOLD["frame_json_position"]) and np.allclose(j["rotation"], GOLD["frame_json_rotation"])
    assert [j["id"], j["width"], j["height"], j["fx"], j["fy"]] == GOLD["frame_json_misc"].tolist() and j["img_name"] == "b"
    got = [S.get_downscale_factor(48, 64, 48, 64), S.get_downscale_factor(48, 64, 24, 32), S.get_downscale_factor(1080, 1920, 675, 1200)]
    assert got == GOLD["downscale"].tolist()
    with pytest.raises(ValueError):
        S.get_downscale_factor(100, 100, 50, 60)
    Image.fromarray(GOLD["rgb"][..., 0], "L").save(tmp_path / "grey.png")
    with pytest.raises(ValueError):
        S.get_image_arr(tmp_path / "grey.png", True)


@pytest.mark.parametrize("model", ["PINHOLE", "SIMPLE_PINHOLE"])
def test_colmap_reader_on_synthetic_model(tmp_path, model):
    truth = write_colmap(tmp_path, n_images=6, n_points=257, model=model, with_masks=True)
    cams = S.load_intrinsics_binary(tmp_path / "sparse" / "0" / "cameras.bin")
    assert list(cams) == [7] and cams[7].model_name == model and (cams[7].width, cams[7].height) == (64, 48)
    p = truth["cameras"][7]["params"]
    exp = (p[0], p[1], p[2], p[3]) if model == "PINHOLE" else (p[0], p[0], p[1], p[2])
    assert (cams[7].fx, cams[7].fy, cams[7].cx, cams[7].cy) == exp
    imgs = S.load_extrinsics_binary(tmp_path / "sparse" / "0" / "images.bin")
    assert sorted(im.image_file_name for im in imgs.values()) == sorted(truth["images"])
    pc = S.load_pointcloud(tmp_path / "sparse" / "0" / "points3D.bin")
    assert pc.nbr_points == 257 and np.array_equal(pc.xyzs, truth["xyzs"]) and np.array_equal(pc.rgbs, truth["rgbs"])
    random.seed(0)
    frames, pc2, train_idx, eval_idx = S.load_colmap_data(str(tmp_path), True, 1, True, 0.34, False)
    assert [f.image_path.name for f in frames] == sorted(truth["images"])          # sorted by path
    assert len(eval_idx) == 2 and len(train_idx) == 4 and sorted(train_idx + eval_idx) == list(range(6))
    for f in frames:
        t = truth["images"][f.image_path.name]
        assert np.allclose(f.w2c, t["w2c"], atol=1e-6) and f.w2c.dtype == np.float32   # un-normalised quats on disk
        assert (f.mask_path is not None) == ("mask" in t)
        d = f.to_data()
        assert np.array_equal((d["image"].numpy() * 255).round().astype(np.uint8), t["image"])
        if "mask" in t:
            assert np.array_equal(d["mask"].numpy().astype(np.uint8), S.expand_mask((t["mask"] > 0).astype(np.uint8), 1))
        else:
            assert float(d["mask"].abs().max()) == 0.0
    _, _, train_all, _ = S.load_colmap_data(str(tmp_path), False, 0, False, 0.34, False)
    assert len(train_all) == 6                                                     # eval=False trains on everything
    with pytest.raises(FileNotFoundError):
        S.load_pointcloud(tmp_path / "nope.bin")


def test_colmap_reader_against_an_independent_spec_literal_writer(tmp_path):
    """the readers were only ever fed by this repository's own writer (tools/make_synthetic_dataset.py).
    Here a second writer, written field by field from COLMAP's documented binary layout and the field order the reference
    consumes (/root/reference/scene/colmap_loader.py:83-160: cameras `iiQQ` + params, images `idddddddi` + a NUL-terminated
    name + `Q` observations of `ddq`, points `QdddBBBd` + `Q` track elements of `ii`), produces what real COLMAP output has
    and the other writer lacks: NON-EMPTY observation / track blocks of varying length that the reader must skip exactly,
    several cameras, ids that are neither dense nor ordered, names of different lengths (with a directory part), and
    un-normalised quaternions.  Every byte is emitted by one explicit struct.pack per field."""
    import struct
    rng = np.random.default_rng(7)
    sparse = tmp_path / "sparse" / "0"
    sparse.mkdir(parents=True)
    # ---- cameras.bin: uint64 count; per camera int32 id, int32 model, uint64 width, uint64 height, float64 params[]
    cams = {11: (1, 640, 480, [500.5, 510.25, 320.0, 240.0]), 3: (1, 800, 600, [700.0, 701.0, 400.5, 299.5])}
    with open(sparse / "cameras.bin", "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, (model, w, h, params) in cams.items():
            f.write(struct.pack("<i", cid)); f.write(struct.pack("<i", model))
            f.write(struct.pack("<Q", w)); f.write(struct.pack("<Q", h))
            for v in params:
                f.write(struct.pack("<d", v))
    # ---- images.bin: uint64 count; per image int32 id, float64 qw qx qy qz, float64 tx ty tz, int32 camera id,
    #      name bytes + NUL, uint64 n observations, then per observation float64 x, float64 y, int64 point3D id
    images = {}
    names = ["a.png", "frames/000123.jpeg", "x" * 70 + ".png", "b.PNG", "c.png"]
    with open(sparse / "images.bin", "wb") as f:
        f.write(struct.pack("<Q", len(names)))
        for k, name in enumerate(names):
            iid = 1000 - 37 * k
            quat = rng.standard_normal(4) * (0.3 + k)          # un-normalised on purpose
            trans = rng.standard_normal(3) * 5
            cid = 11 if k % 2 else 3
            n_obs = int(rng.integers(0, 40))
            f.write(struct.pack("<i", iid))
            for v in quat:
                f.write(struct.pack("<d", float(v)))
            for v in trans:
                f.write(struct.pack("<d", float(v)))
            f.write(struct.pack("<i", cid))
            f.write(name.encode("utf-8")); f.write(b"\x00")
            f.write(struct.pack("<Q", n_obs))
            for _ in range(n_obs):
                f.write(struct.pack("<d", float(rng.random() * 640))); f.write(struct.pack("<d", float(rng.random() * 480)))
                f.write(struct.pack("<q", int(rng.integers(-1, 10**6))))
            images[iid] = (name, cid, quat, trans)
    # ---- points3D.bin: uint64 count; per point uint64 id, float64 xyz, uint8 rgb, float64 error, uint64 track length,
    #      then per track element int32 image id, int32 point2D index
    n_pts = 301
    xyz = rng.standard_normal((n_pts, 3)) * 3
    rgb = rng.integers(0, 256, (n_pts, 3))
    with open(sparse / "points3D.bin", "wb") as f:
        f.write(struct.pack("<Q", n_pts))
        for i in range(n_pts):
            f.write(struct.pack("<Q", 5 * i + 2))
            for v in xyz[i]:
                f.write(struct.pack("<d", float(v)))
            for v in rgb[i]:
                f.write(struct.pack("<B", int(v)))
            f.write(struct.pack("<d", float(rng.random())))
            n_tr = int(rng.integers(0, 9))
            f.write(struct.pack("<Q", n_tr))
            for _ in range(n_tr):
                f.write(struct.pack("<i", int(rng.integers(1, 1000)))); f.write(struct.pack("<i", int(rng.integers(0, 5000))))
    got_c = S.load_intrinsics_binary(sparse / "cameras.bin")
    assert set(got_c) == {3, 11}
    for cid, (model, w, h, params) in cams.items():
        c = got_c[cid]
        assert c.model_name == "PINHOLE" and (c.width, c.height) == (w, h) and (c.fx, c.fy, c.cx, c.cy) == tuple(params)
    got_i = S.load_extrinsics_binary(sparse / "images.bin")
    assert set(got_i) == set(images)
    for iid, (name, cid, quat, trans) in images.items():
        im = got_i[iid]
        assert im.image_file_name == name and im.camera_id == cid
        assert tuple(im.quat) == tuple(float(v) for v in quat) and tuple(im.trans) == tuple(float(v) for v in trans)   # bit-exact doubles
    pc = S.load_pointcloud(sparse / "points3D.bin")
    assert pc.nbr_points == n_pts and pc.xyzs.dtype == np.float32 and pc.rgbs.dtype == np.uint8
    assert np.array_equal(pc.xyzs, xyz.astype(np.float32)) and np.array_equal(pc.rgbs, rgb.astype(np.uint8))
    # frames: w2c = [R(q / |q|) | t] (pyquaternion normalises, /root/reference/scene/colmap_loader.py:175-176), per-image camera
    frames, _, train_idx, eval_idx = S.load_colmap_data(str(tmp_path), False, 0, False, 0.0, False)
    assert [f.image_path.name for f in frames] == [Path(n).name for n in sorted(names, key=lambda n: str(tmp_path / "images" / n))]
    by_name = {str(Path("images") / v[0]): v for v in images.values()}
    for fr in frames:
        name, cid, quat, trans = by_name[str(fr.image_path.relative_to(tmp_path))]
        q = quat / np.linalg.norm(quat)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        assert np.allclose(fr.w2c[:3, :3], R, atol=1e-6) and np.allclose(fr.w2c[:3, 3], trans.astype(np.float32)) and fr.w2c[3].tolist() == [0, 0, 0, 1]
        assert (fr.width, fr.height, fr.fx, fr.fy, fr.cx, fr.cy) == (cams[cid][1], cams[cid][2], *cams[cid][3])
    assert sorted(train_idx) == list(range(5)) and eval_idx == []
    # a truncated file must raise, not return garbage
    raw = (sparse / "images.bin").read_bytes()
    (sparse / "images.bin").write_bytes(raw[:-5])
    with pytest.raises(Exception):
        S.load_extrinsics_binary(sparse / "images.bin")


def test_colmap_downscaled_images_rescale_intrinsics(tmp_path):
    truth = write_colmap(tmp_path, n_images=2, width=64, height=48, image_scale=0.5, with_masks=False)
    frames, _, _, _ = S.load_colmap_data(str(tmp_path), False, 0, False, 0.0, True)
    d = frames[0].to_data()
    p = truth["cameras"][7]["params"]
    assert (d["height"], d["width"]) == (24, 32)
    assert np.allclose(d["K"].numpy(), [[p[0] / 2, 0, p[2] / 2], [0, p[1] / 2, p[3] / 2], [0, 0, 1]])


def test_blender_reader_on_synthetic_scene(tmp_path):
    truth = write_blender(tmp_path, n_train=4, n_val=2, n_test=3, size=40)
    np.random.seed(0)
    frames, pc, train_idx, eval_idx = S.load_blender_data(str(tmp_path), True, 0, True, False, True, True)
    assert len(frames) == 7 and eval_idx == [0, 1, 2] and train_idx == [3, 4, 5, 6]   # eval (test) frames first
    order = truth["splits"]["test"] + truth["splits"]["train"]
    fx = 40 / (2 * np.tan(truth["camera_angle_x"] / 2))
    for f, t in zip(frames, order):
        c2w = t["c2w_blender"].copy()
        c2w[:3, 1:3] *= -1                                                            # OpenGL -> OpenCV axes
        assert np.allclose(f.w2c, np.linalg.inv(c2w)) and (f.width, f.height) == (40, 40)
        assert abs(f.fx - fx) < 1e-9 and f.fx == f.fy and (f.cx, f.cy) == (20.0, 20.0)
        rgba = t["image_rgba"].astype(np.float64)
        a = rgba[..., 3:4] / 255.0
        assert np.array_equal(S.get_image_arr(f.image_path, True), (rgba[..., :3] * a + 255.0 * (1 - a)).astype(np.uint8))
    assert sum(f.mask_path is not None for f in frames) == 1
    assert pc.nbr_points == 100000 and pc.rgbs.dtype == np.uint8 and int(pc.rgbs.max()) == 127
    cams = np.stack([np.linalg.inv(f.w2c)[:3, 3] for f in frames[3:]])
    lo, hi = cams.min(), cams.max()
    c = (lo + hi) / 2
    assert pc.xyzs.min() >= c - (c - lo) / 3 - 1e-9 and pc.xyzs.max() <= c + (hi - c) / 3 + 1e-9
    frames2, _, train2, eval2 = S.load_blender_data(str(tmp_path), False, 0, False, True, True, False)
    assert len(frames2) == 9 and eval2 == [0, 1, 2, 3, 4] and train2 == list(range(9))


def test_scene_cycles_training_frames_and_exports_cameras(tmp_path):
    write_colmap(tmp_path / "data", n_images=5, with_masks=False)
    (tmp_path / "out").mkdir()
    random.seed(1)
    sc = S.Scene(str(tmp_path / "data"), "colmap", str(tmp_path / "out"), 12, True, 0.2, False, False, False, 0, False)
    assert sc.nbr_data("train") == 12 and sc.nbr_data("eval") == 1 and len(sc.train_dataset) == 12
    assert sc.train_indexes[:4] == sc.train_indexes[4:8]                              # the list repeats to fill the iterations
    d = sc.train_dataset[0]
    assert set(d) == {"K", "height", "width", "w2c", "image", "mask"}
    states = ckpt.load_camera_states(tmp_path / "out")
    assert len(states) == 5
    for st, fr in zip(states, sc.frames):
        assert np.allclose(st.w2c, fr.w2c, atol=1e-5) and st.K[0, 0] == np.float32(fr.fx) and st.K[0, 2] == fr.width / 2
    with pytest.raises(ValueError):
        S.Scene(str(tmp_path / "data"), "colmap", None, 2, False, 0.0, False, False, False, 0, False)   # fewer iterations than frames
    with pytest.raises(ValueError):
        S.Scene(str(tmp_path / "data"), "ply", None, 10, False, 0.0, False, False, False, 0, False)


def _small_model(n=40):
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    return GaussianModel(means=r(n, 3), log_scales=r(n, 3) * 0.1 - 3, quats=r(n, 4), sh_0=r(n, 1, 3), sh_rest=r(n, 15, 3), logit_opacities=r(n),
                         sh_degree=3, sh_degree_interval=1000, white_background=True)


def test_checkpoint_layout_selection_and_reference_class_paths(tmp_path):
    m = _small_model()
    opt = build_optimizers(m, 1.6e-4, 5e-3, 1e-3, 2.5e-3, 1.25e-4, 5e-2)
    for it in (7000, 30000, 15000):
        with torch.no_grad():
            m.means.add_(1.0)
        ckpt.save_gaussian_model(tmp_path / "checkpoints" / f"iterations_{it}.pth", m)
    assert m.optimizer is opt                                                         # detached only while saving
    assert ckpt.find_checkpoint(tmp_path).name == "iterations_30000.pth"
    assert ckpt.find_checkpoint(tmp_path, 7000).name == "iterations_7000.pth"
    with pytest.raises(ValueError):
        ckpt.find_checkpoint(tmp_path, 1)
    with pytest.raises(ValueError):
        ckpt.find_checkpoint(tmp_path / "empty")
    latest = ckpt.load_gaussian_model(tmp_path, device="cpu")
    first = ckpt.load_gaussian_model(tmp_path, 7000, device="cpu")
    assert isinstance(latest, GaussianModel) and latest.optimizer is None
    assert torch.allclose(latest.means, m.means - 1.0, atol=1e-5) and torch.allclose(first.means, m.means - 2.0, atol=1e-5)   # saved after 2 / 1 of the 3 increments
    assert latest.active_sh_degree == 0 and latest.MAX_SH_DEGREE == 3 and latest.BACKGROUND.tolist() == [1.0, 1.0, 1.0]
    assert latest.means_lr_scheduler(0) == m.means_lr_scheduler(0)
    # the pickle names the reference's class paths, so the reference's own torch.load resolves it to ITS classes
    raw = (tmp_path / "checkpoints" / "iterations_7000.pth").read_bytes()
    import zipfile
    with zipfile.ZipFile(tmp_path / "checkpoints" / "iterations_7000.pth") as z:
        pkl = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    assert b"model.gaussian" in pkl and b"GaussianModel" in pkl and b"model.utils" in pkl and b"easy_gaussian_splatting_amd.model" not in pkl
    assert "model" not in sys.modules and GaussianModel.__module__ == "easy_gaussian_splatting_amd.model"   # aliases removed
    # with the optimizer
    ckpt.save_gaussian_model(tmp_path / "checkpoints" / "iterations_40000.pth", m, save_optimizer=True)
    full = ckpt.load_gaussian_model(tmp_path, device="cpu")
    assert full.optimizer is not None and [g["name"] for g in full.optimizer.param_groups] == m.param_names


def test_checkpoint_written_by_the_reference_loads_renders_and_resumes(tmp_path):
    """A file as the REFERENCE writes it: an object of class `model.gaussian.GaussianModel` that carries exactly the
    reference's attribute set (/root/reference/model/gaussian.py:31-92: statistics as plain tensor attributes,
    MAX_SCALE_RATIO a 0-d tensor, a torch.optim.Adam inside, none of this package's extra attributes).  It must unpickle
    into this package's class, fall back to the class defaults (`fuse_sh_cat`), and its optimizer must step."""
    import types
    n = 30
    g = torch.Generator().manual_seed(1)
    r = lambda *sh: torch.randn(*sh, generator=g)
    # stand-in classes under the reference's module paths, with the reference's attribute names only
    pkg, gmod, umod = types.ModuleType("model"), types.ModuleType("model.gaussian"), types.ModuleType("model.utils")
    pkg.__path__ = []

    class LR_Scheduler:   # noqa: N801  (the reference's name)
        def __init__(self, lr_init, lr_final, max_steps):
            self.lr_init, self.lr_final, self.max_steps = lr_init, lr_final, max_steps

    class RefGaussianModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            P = torch.nn.Parameter
            self.means, self.log_scales, self.quats = P(r(n, 3)), P(r(n, 3) * 0.1 - 3), P(r(n, 4))
            self.sh_0, self.sh_rest, self.logit_opacities = P(r(n, 1, 3)), P(r(n, 15, 3)), P(r(n))
            self.grad_norm_accum, self.collecting_counts, self.max_radii = torch.zeros(n), torch.zeros(n), torch.zeros(n)
            self.optimizer = None
            self.active_sh_degree = 2
            self.means_lr_scheduler = LR_Scheduler(1.6e-4, 1.6e-6, 30000)
            self.MAX_SH_DEGREE = 3
            self.DENSIFY_GRAD_THRESH, self.DENSIFY_SCALE_THRESH, self.NUM_SPLITS = 0.0002, 0.01, 2
            self.PRUNE_RADII_RATIO_THRESH, self.PRUNE_SCALE_THRESH, self.MIN_OPACITY = 0.15, 0.1, 0.005
            self.USE_SCALE_REGULARIZATION = True
            self.MAX_SCALE_RATIO = torch.tensor(10.0)
            self.BACKGROUND = P(torch.zeros(3), requires_grad=False)

    LR_Scheduler.__module__, LR_Scheduler.__qualname__ = "model.utils", "LR_Scheduler"
    RefGaussianModel.__module__, RefGaussianModel.__qualname__, RefGaussianModel.__name__ = "model.gaussian", "GaussianModel", "GaussianModel"
    gmod.GaussianModel, umod.LR_Scheduler = RefGaussianModel, LR_Scheduler
    ref = RefGaussianModel()
    names = ["means", "log_scales", "quats", "sh_0", "sh_rest", "logit_opacities"]
    ref.optimizer = torch.optim.Adam([{"params": [getattr(ref, k)], "lr": 1e-3 * (i + 1), "name": k} for i, k in enumerate(names)])
    for k in names:
        getattr(ref, k).grad = torch.ones_like(getattr(ref, k))
    ref.optimizer.step()
    sys.modules.update({"model": pkg, "model.gaussian": gmod, "model.utils": umod})
    try:
        (tmp_path / "checkpoints").mkdir()
        torch.save(ref, tmp_path / "checkpoints" / "iterations_100.pth")
    finally:
        for k in ("model", "model.gaussian", "model.utils"):
            sys.modules.pop(k)
    m = ckpt.load_gaussian_model(tmp_path, device="cpu")
    assert type(m) is GaussianModel and "fuse_sh_cat" not in m.__dict__ and m.fuse_sh_cat is True and m.tile_culling == "tight"
    assert torch.equal(m.means, ref.means) and m.active_sh_degree == 2 and isinstance(m.max_radii, torch.Tensor)
    assert abs(m.means_lr_scheduler(15000) - 1.6e-5) < 1e-9                      # this package's __call__ on the reference's fields
    assert float(m.get_regularization_dict()["scale_reg"]) >= 0.0                # MAX_SCALE_RATIO is a tensor in such files
    assert isinstance(m.optimizer, torch.optim.Adam) and [g["name"] for g in m.optimizer.param_groups] == names
    before = m.means.detach().clone()
    for k in names:
        getattr(m, k).grad = torch.ones_like(getattr(m, k))
    m.optimizer.step()                                                           # the pickled state resumes (step 2 of Adam)
    st = m.optimizer.state[m.means]
    assert float(st["step"]) == 2.0 and not torch.equal(m.means.detach(), before)
    # and the way back: what this package writes holds only what the reference's classes can hold
    m.USE_SCALE_REGULARIZATION, m.MAX_SCALE_RATIO = True, 10.0
    ckpt.save_gaussian_model(tmp_path / "checkpoints" / "iterations_200.pth", m, save_optimizer=True)
    assert m.MAX_SCALE_RATIO == 10.0 and isinstance(m.optimizer, torch.optim.Adam)
    import zipfile
    with zipfile.ZipFile(tmp_path / "checkpoints" / "iterations_200.pth") as z:
        pkl = z.read([nm for nm in z.namelist() if nm.endswith("data.pkl")][0])
    assert b"easy_gaussian_splatting_amd" not in pkl                              # no class of this package is named
    with ckpt.reference_class_paths():
        raw = torch.load(tmp_path / "checkpoints" / "iterations_200.pth", map_location="cpu", weights_only=False)
    assert isinstance(raw.MAX_SCALE_RATIO, torch.Tensor) and float(torch.max(torch.tensor([3.0, 30.0]), raw.MAX_SCALE_RATIO)[0]) == 10.0


def test_model_from_pointcloud_follows_reference_initialisation():
    rng = np.random.default_rng(0)
    pc = S.Pointcloud(rng.standard_normal((200, 3)).astype(np.float32), rng.integers(0, 256, (200, 3), dtype=np.uint8))
    m = GaussianModel.from_pointcloud(pc, sh_degree=3, sh_degree_interval=1000, white_background=False)
    assert m.nbr_gaussians == 200 and m.sh_rest.shape == (200, 15, 3) and float(m.sh_rest.detach().abs().max()) == 0.0
    assert torch.allclose(m.opacities, torch.full((200,), 0.8)) and torch.equal(m.quats[:, 0], torch.ones(200))
    d = np.linalg.norm(pc.xyzs[:, None] - pc.xyzs[None], axis=-1)
    d.sort(axis=1)
    exp = d[:, 1:4].mean(1) / 2.0
    assert np.allclose(m.scales.detach().numpy(), np.repeat(exp[:, None], 3, 1), rtol=1e-5)
    assert np.allclose(m.sh_0[:, 0].detach().numpy(), (pc.rgbs / 255.0 - 0.5) / 0.28209479177387814, atol=1e-6)
    assert m.active_sh_degree == 0


def test_loader_quaternion_convention_matches_reference_fixture():
    """The COLMAP reader's own wxyz -> rotation routine against vectors captured from the reference's
    model/utils.py:31-55 (the convention pyquaternion's `rotation_matrix` shares), un-normalised inputs included."""
    z = np.load(Path(__file__).parent / "golden" / "ref_model_utils.npz")
    for q, R in zip(z["quats"], z["rotmats"]):
        assert np.allclose(S.quat_wxyz_to_rotmat(q), R, atol=1e-12)
