#!/usr/bin/env python3
"""Writes small synthetic datasets in the two on-disk formats the reference reads (no real capture exists here):

  colmap:   <root>/sparse/0/{cameras,images,points3D}.bin  + images/*.png [+ masks/*.png]
            (COLMAP's binary model files, the layout /root/reference/scene/colmap_loader.py:83-149 parses)
  blender:  <root>/transforms_{train,val,test}.json + {train,val,test}/r_*.png (RGBA) [+ train_masks/r_*.png]
            (nerf_synthetic, /root/reference/scene/blender_loader.py:10-57)

and returns the ground truth it wrote, so tests can compare what the readers return with what went in.
    python tools/make_synthetic_dataset.py <out_dir> [colmap|blender]
"""
from __future__ import annotations

import json
import struct
import sys
from pathlib import Path

import numpy as np
from PIL import Image


def _rot(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return q, R


def write_colmap(root: Path, n_images=5, n_points=300, width=64, height=48, seed=0, model="PINHOLE", with_masks=True,
                 image_scale=1.0, unnormalised_quats=True):
    rng = np.random.default_rng(seed)
    sparse = root / "sparse" / "0"
    sparse.mkdir(parents=True, exist_ok=True)
    (root / "images").mkdir(exist_ok=True)
    if with_masks:
        (root / "masks").mkdir(exist_ok=True)
    truth = {"cameras": {}, "images": {}, "xyzs": None, "rgbs": None}
    # cameras.bin: u64 count | per camera: i32 id, i32 model, u64 w, u64 h, f64 params[]
    cams = {7: (model, [55.5, 54.25, width / 2 + 0.5, height / 2 - 0.25] if model == "PINHOLE" else [57.0, width / 2, height / 2])}
    with open(sparse / "cameras.bin", "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, (name, params) in cams.items():
            f.write(struct.pack("<iiQQ", cid, {"SIMPLE_PINHOLE": 0, "PINHOLE": 1}[name], width, height))
            f.write(struct.pack("<" + "d" * len(params), *params))
            truth["cameras"][cid] = dict(model=name, params=params, width=width, height=height)
    # images.bin: u64 count | per image: i32 id, 4 f64 quat (wxyz), 3 f64 t, i32 camera, name\0, u64 n2d, n2d x (f64, f64, i64)
    names = [f"frame_{(7 * i) % n_images:03d}.png" for i in range(n_images)]   # not in sorted order on purpose
    with open(sparse / "images.bin", "wb") as f:
        f.write(struct.pack("<Q", n_images))
        for i, name in enumerate(names):
            q, R = _rot(rng)
            qs = q * (1.0 + 0.3 * i) if unnormalised_quats else q    # readers must normalise
            t = rng.standard_normal(3)
            f.write(struct.pack("<idddddddi", 100 + i, *qs, *t, 7))
            f.write(name.encode("utf-8") + b"\x00")
            n2d = int(rng.integers(0, 6))
            f.write(struct.pack("<Q", n2d))
            for _ in range(n2d):
                f.write(struct.pack("<ddq", rng.random(), rng.random(), int(rng.integers(-1, 50))))
            w2c = np.eye(4)
            w2c[:3, :3], w2c[:3, 3] = R, t
            truth["images"][name] = dict(id=100 + i, w2c=w2c)
            ih, iw = int(round(height * image_scale)), int(round(width * image_scale))
            img = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
            Image.fromarray(img, "RGB").save(root / "images" / name)
            truth["images"][name]["image"] = img
            if with_masks and i % 2 == 0:
                m = (rng.random((ih, iw)) > 0.97).astype(np.uint8) * 255
                Image.fromarray(m, "L").save((root / "masks" / name).with_suffix(".png"))
                truth["images"][name]["mask"] = m
    # points3D.bin: u64 count | per point: u64 id, 3 f64 xyz, 3 u8 rgb, f64 error, u64 track, track x (i32, i32)
    xyzs = rng.standard_normal((n_points, 3))
    rgbs = rng.integers(0, 256, (n_points, 3), dtype=np.uint8)
    with open(sparse / "points3D.bin", "wb") as f:
        f.write(struct.pack("<Q", n_points))
        for i in range(n_points):
            f.write(struct.pack("<QdddBBBd", 1000 + i, *xyzs[i], *[int(c) for c in rgbs[i]], float(rng.random())))
            track = int(rng.integers(0, 5))
            f.write(struct.pack("<Q", track))
            for _ in range(track):
                f.write(struct.pack("<ii", int(rng.integers(0, 100)), int(rng.integers(0, 100))))
    truth["xyzs"], truth["rgbs"] = xyzs.astype(np.float32), rgbs
    return truth


def write_blender(root: Path, n_train=4, n_val=2, n_test=3, size=40, seed=1, with_masks=True):
    rng = np.random.default_rng(seed)
    root.mkdir(parents=True, exist_ok=True)
    fov = 0.6911112070083618
    truth = {"camera_angle_x": fov, "splits": {}}
    for split, n in (("train", n_train), ("val", n_val), ("test", n_test)):
        (root / split).mkdir(exist_ok=True)
        frames, tl = [], []
        for i in range(n):
            _, R = _rot(rng)
            c2w = np.eye(4)
            c2w[:3, :3], c2w[:3, 3] = R, rng.standard_normal(3) * 2
            frames.append({"file_path": f"./{split}/r_{i}", "rotation": 0.1, "transform_matrix": c2w.tolist()})
            img = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
            Image.fromarray(img, "RGBA").save(root / split / f"r_{i}.png")
            entry = dict(c2w_blender=c2w, image_rgba=img)
            if with_masks and split == "train" and i == 1:
                (root / "train_masks").mkdir(exist_ok=True)
                m = (rng.random((size, size)) > 0.95).astype(np.uint8) * 200
                Image.fromarray(m, "L").save(root / "train_masks" / f"r_{i}.png")
                entry["mask"] = m
            tl.append(entry)
        with open(root / f"transforms_{split}.json", "w") as f:
            json.dump({"camera_angle_x": fov, "frames": frames}, f)
        truth["splits"][split] = tl
    return truth


if __name__ == "__main__":
    out = Path(sys.argv[1])
    kind = sys.argv[2] if len(sys.argv) > 2 else "colmap"
    t = write_colmap(out) if kind == "colmap" else write_blender(out)
    print(f"wrote a synthetic {kind} dataset under {out}")
