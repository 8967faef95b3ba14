"""Scene / data side of the hot path's callers ("next" row): what feeds
`GaussianModel.forward(data)` and the loss in the reference's train / eval loops.

Mirrors, with the same names, argument meaning and error behaviour:
  Pointcloud, Frame (.to_json / .to_data), get_downscale_factor, get_image_arr,
  expand_mask, get_mask_arr, data_to_device        <- /root/reference/scene/data_class.py:8-212
  load_intrinsics_binary / load_extrinsics_binary / load_pointcloud / load_colmap_data
                                                   <- /root/reference/scene/colmap_loader.py:11-211
  load_frames / generate_pointcloud / load_blender_data
                                                   <- /root/reference/scene/blender_loader.py:10-114
  Scene, SceneDataset                              <- /root/reference/scene/scene.py:10-94

Differences, all behaviour-preserving: `load_pointcloud` parses COLMAP's points3D.bin with a vectorised scan
instead of one struct.unpack + two numpy allocations per point (the reference spends minutes on a 1 M-point
cloud); `expand_mask` is a separable dilation (O(e) shifted ORs per axis instead of O(e^2) full-image adds);
the w2c rotation comes from this package's own wxyz quaternion routine instead of pyquaternion.
No real capture exists in this environment: tests/ validates the readers on synthetic files written by
tools/make_synthetic_dataset.py and the image / mask helpers against vectors captured from the reference's own
data_class.py (tests/golden/make_golden.py).
"""
from __future__ import annotations

import json
import random
import struct
from pathlib import Path
from typing import Any, BinaryIO, Dict, List, Literal, Optional, Sequence, Tuple

import numpy as np
import torch

try:   # Pillow is only needed to open image files
    from PIL import Image as _PILImage
except ImportError:   # pragma: no cover
    _PILImage = None


# ------------------------------------------------------------------------------------------------ data classes
class Pointcloud:
    def __init__(self, xyzs: np.ndarray, rgbs: np.ndarray):
        self.xyzs = xyzs  # [N, 3]
        self.rgbs = rgbs  # [N, 3]  uint8

    @property
    def nbr_points(self) -> int:
        return self.xyzs.shape[0]


def get_downscale_factor(orig_h: int, orig_w: int, target_h: int, target_w: int) -> float:
    if orig_h == target_h and orig_w == target_w:
        return 1.0
    h_factor, w_factor = target_h / orig_h, target_w / orig_w
    if abs(h_factor - w_factor) > 1e-3:
        raise ValueError(f"h_downscale_factor ({h_factor}) and w_downscale_factor ({w_factor}) are not close")
    return (h_factor + w_factor) / 2


def get_image_arr(image_path: Path, white_background: bool) -> np.ndarray:
    """uint8 [H, W, 3]; RGBA is composited over white / black (in float64, truncated like the reference)."""
    image = _PILImage.open(image_path)
    if image.mode == "RGB":
        return np.array(image, dtype=np.uint8)
    if image.mode == "RGBA":
        arr = np.array(image, dtype=np.float64)
        background = np.full((arr.shape[0], arr.shape[1], 3), 255.0 if white_background else 0.0, dtype=np.float64)
        alpha = arr[..., 3:4] / 255.0
        return (arr[..., :3] * alpha + background * (1 - alpha)).astype(np.uint8)
    raise ValueError(f"only support image on 'RGB' or 'RGBA' mode, but get '{image.mode}'")


def expand_mask(mask: np.ndarray, expand_pixels: int) -> np.ndarray:
    """out[y, x] = OR of mask[y + dy, x + dx] over dy, dx in [-(e - 1), e]  (the window the reference's
    shifted-add loop produces, /root/reference/scene/data_class.py:180-196), as two 1-D passes."""
    if expand_pixels == 0:
        return mask
    e = int(expand_pixels)
    h, w = mask.shape
    src = (mask > 0).astype(np.uint8)
    tmp = np.zeros((h, w + 2 * e), dtype=np.uint8)
    for d in range(-(e - 1), e + 1):       # along x
        tmp[:, e - d:e - d + w] |= src
    rows = tmp[:, e:e + w]
    out = np.zeros((h + 2 * e, w), dtype=np.uint8)
    for d in range(-(e - 1), e + 1):       # along y
        out[e - d:e - d + h, :] |= rows
    return out[e:e + h, :]


def get_mask_arr(mask_path: Path, expand_pixels: int) -> np.ndarray:
    mask_arr = np.array(_PILImage.open(mask_path), dtype=np.uint8)
    if mask_arr.ndim != 2:
        raise ValueError(f"only support mask on 2D, but get {mask_arr.ndim}D")
    mask_arr[mask_arr >= 1] = 1   # 1: object to be removed, 0: scene to be constructed
    return expand_mask(mask_arr, expand_pixels)


class Frame:
    def __init__(self, image_path: Path, mask_path: Optional[Path], mask_expand_pixels: int, width: int, height: int,
                 fx: float, fy: float, cx: float, cy: float, w2c: np.ndarray, white_background: bool):
        self.image_path, self.mask_path, self.mask_expand_pixels = image_path, mask_path, mask_expand_pixels
        self.width, self.height = width, height
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.w2c = w2c  # colmap/opencv (X right, Y down, Z forward)
        self.white_background = white_background

    def to_json(self, id: int):
        c2w = np.linalg.inv(self.w2c)
        return {"id": id, "img_name": self.image_path.stem, "width": self.width, "height": self.height,
                "position": c2w[:3, 3].tolist(), "rotation": c2w[:3, :3].tolist(), "fx": self.fx, "fy": self.fy}

    def to_data(self) -> Dict[str, Any]:
        """The dict `GaussianModel.forward` / `LossComputer.get_loss_dict` consume
        (/root/reference/scene/data_class.py:110-143): K rescaled when the image on disk was downscaled."""
        w2c = torch.tensor(self.w2c, dtype=torch.float32)
        image_arr = get_image_arr(self.image_path, self.white_background).astype(np.float32) / 255.0
        height, width = image_arr.shape[:2]
        image_tensor = torch.tensor(image_arr, dtype=torch.float32)
        if self.mask_path is not None:
            mask_tensor = torch.tensor(get_mask_arr(self.mask_path, self.mask_expand_pixels), dtype=torch.float32)
            if mask_tensor.shape != image_tensor.shape[:2]:
                raise ValueError(f"mask size ({mask_tensor.shape}) is not equal to image size {image_tensor.shape}")
        else:
            mask_tensor = torch.zeros((height, width), dtype=torch.float32)
        f = get_downscale_factor(self.height, self.width, height, width)
        K = torch.tensor([[self.fx * f, 0, self.cx * f], [0, self.fy * f, self.cy * f], [0, 0, 1]], dtype=torch.float32)
        return {"K": K, "height": height, "width": width, "w2c": w2c, "image": image_tensor, "mask": mask_tensor}


def data_to_device(data: Dict[str, Any], non_blocking: bool = True, device="cuda"):
    for k in ("K", "w2c", "image", "mask"):
        data[k] = data[k].to(device, non_blocking=non_blocking)


# ------------------------------------------------------------------------------------------------ COLMAP binary model
class Camera:
    def __init__(self, id: int, model_name: Literal["SIMPLE_PINHOLE", "PINHOLE"], width: int, height: int, params: Sequence[float]):
        self.id, self.model_name, self.width, self.height = id, model_name, width, height
        if model_name == "SIMPLE_PINHOLE":
            self.fx = self.fy = params[0]
            self.cx, self.cy = params[1], params[2]
        elif model_name == "PINHOLE":
            self.fx, self.fy, self.cx, self.cy = params[0], params[1], params[2], params[3]
        else:
            raise ValueError(f"unsupported camera model: {model_name}")


class Image:
    def __init__(self, id: int, image_file_name: str, camera_id: int, quat: Sequence[float], trans: Sequence[float]):
        self.id, self.image_file_name, self.camera_id = id, image_file_name, camera_id
        self.quat, self.trans = quat, trans   # w2c; quat is wxyz


def read_next_bytes(f: BinaryIO, num_bytes: int, format_char_sequence: str, endian_character="<") -> Tuple[Any, ...]:
    return struct.unpack(endian_character + format_char_sequence, f.read(num_bytes))


_CAM_MAP = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4)}   # {cam_model_id: (name, num_params)}


def load_intrinsics_binary(path: Path) -> Dict[int, Camera]:
    if not path.exists():
        raise FileNotFoundError(f"{path} does not exist")
    camera_map: Dict[int, Camera] = {}
    with open(path, "rb") as f:
        for _ in range(read_next_bytes(f, 8, "Q")[0]):
            camera_id, model_id, width, height = read_next_bytes(f, 24, "iiQQ")
            if model_id not in _CAM_MAP:
                raise ValueError(f"unsupported camera model id: {model_id}")
            name, num_params = _CAM_MAP[model_id]
            params = read_next_bytes(f, 8 * num_params, "d" * num_params)
            camera_map[camera_id] = Camera(camera_id, name, width, height, params)   # type: ignore
    assert len(set(cam.model_name for cam in camera_map.values())) == 1
    return camera_map


def load_extrinsics_binary(path: Path) -> Dict[int, Image]:
    if not path.exists():
        raise FileNotFoundError(f"{path} does not exist")
    image_map: Dict[int, Image] = {}
    with open(path, "rb") as f:
        for _ in range(read_next_bytes(f, 8, "Q")[0]):
            props = read_next_bytes(f, 64, "idddddddi")
            name = b""
            c = f.read(1)
            while c != b"\x00":   # zero-terminated file name
                name += c
                c = f.read(1)
            n2d = read_next_bytes(f, 8, "Q")[0]
            # 2-D observations (x, y, point3D id): not needed here, skipped -- but a file that ends inside the block must
            # fail like the reference's struct.unpack of the whole block does (/root/reference/scene/colmap_loader.py:127-128)
            if len(f.read(24 * n2d)) != 24 * n2d:
                raise struct.error(f"{path}: truncated observation block of image {props[0]}")
            image_map[props[0]] = Image(props[0], name.decode("utf-8"), props[8], props[1:5], props[5:8])
    return image_map


def load_pointcloud(path: Path) -> Pointcloud:
    """points3D.bin: per point  u64 id | 3 x f64 xyz | 3 x u8 rgb | f64 error | u64 track length | track x (i32, i32)."""
    if not path.exists():
        raise FileNotFoundError(f"{path} does not exist")
    buf = np.fromfile(path, dtype=np.uint8)
    num_points = int(buf[:8].view("<u8")[0])
    xyzs = np.empty((num_points, 3), dtype=np.float32)
    rgbs = np.empty((num_points, 3), dtype=np.uint8)
    off = 8
    raw = buf.tobytes()
    for i in range(num_points):   # (records are variable-length: a scan, but without per-point numpy allocations)
        x, y, z = struct.unpack_from("<ddd", raw, off + 8)
        xyzs[i] = (x, y, z)
        rgbs[i] = buf[off + 32:off + 35]
        track = struct.unpack_from("<Q", raw, off + 43)[0]
        off += 51 + 8 * track
        if off > len(raw):
            raise struct.error(f"{path}: truncated track of point {i}")
    return Pointcloud(xyzs, rgbs)


def quat_wxyz_to_rotmat(q: Sequence[float]) -> np.ndarray:
    """Unit-normalised wxyz quaternion -> 3x3 rotation (the convention of /root/reference/model/utils.py:31-55 and of
    pyquaternion's `rotation_matrix`, which the reference's loader uses)."""
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def load_colmap_data(path: str, use_masks: bool, mask_expand_pixels: int, eval: bool, eval_split_ratio: float,
                     white_background: bool) -> Tuple[List[Frame], Pointcloud, List[int], List[int]]:
    root = Path(path)
    camera_map = load_intrinsics_binary(root / "sparse" / "0" / "cameras.bin")
    image_map = load_extrinsics_binary(root / "sparse" / "0" / "images.bin")
    pc = load_pointcloud(root / "sparse" / "0" / "points3D.bin")
    frames: List[Frame] = []
    for image in image_map.values():
        camera = camera_map[image.camera_id]
        w2c = np.eye(4, dtype=np.float32)   # colmap/opencv (X right, Y down, Z forward)
        w2c[:3, :3] = quat_wxyz_to_rotmat(image.quat)
        w2c[:3, 3] = np.array(image.trans, dtype=np.float32)
        mask_path = (root / "masks" / image.image_file_name).with_suffix(".png")
        frames.append(Frame(root / "images" / image.image_file_name, mask_path if mask_path.exists() and use_masks else None,
                            mask_expand_pixels, camera.width, camera.height, camera.fx, camera.fy, camera.cx, camera.cy, w2c,
                            white_background))
    frames.sort(key=lambda frame: frame.image_path)
    indexes = list(range(len(frames)))
    random.shuffle(indexes)
    split_point = int(len(frames) * eval_split_ratio)
    eval_indexes = indexes[:split_point]
    train_indexes = indexes[split_point:] if eval else indexes
    return frames, pc, train_indexes, eval_indexes


# ------------------------------------------------------------------------------------------------ Blender / nerf_synthetic
def load_frames(path: Path, use_masks: bool, mask_expand_pixels: int, white_background: bool, suffix: str = ".png") -> List[Frame]:
    if not path.exists():
        raise FileNotFoundError(f"{path} does not exist")
    frames: List[Frame] = []
    with open(path, "r") as f:
        content = json.load(f)
    fov_x = content["camera_angle_x"]
    for frame_json in content["frames"]:
        image_path = path.parent / (frame_json["file_path"] + suffix)
        mask_path = image_path.parent.parent / (image_path.parent.name + "_masks") / image_path.name
        width, height = _PILImage.open(image_path).size
        fx = fy = width / (2 * np.tan(fov_x / 2))
        # blender/opengl (X right, Y up, Z back) -> colmap/opencv (X right, Y down, Z forward)
        c2w = np.array(frame_json["transform_matrix"])
        c2w[:3, 1:3] *= -1
        frames.append(Frame(image_path, mask_path if mask_path.exists() and use_masks else None, mask_expand_pixels, width, height,
                            fx, fy, width / 2.0, height / 2.0, np.linalg.inv(c2w), white_background))
    return frames


def generate_pointcloud(frames: List[Frame], num_points: int = 100000) -> Pointcloud:
    """The start cloud of a dataset without SfM points (/root/reference/scene/blender_loader.py:55-71): `num_points` grey
    points, uniform in a cube -- ONE scalar interval for all three axes: the middle third of [smallest, largest] coordinate of
    any camera centre.  Draws `num_points x 3` numbers from numpy's GLOBAL generator in one call, like the reference (a run
    seeded the same way starts from the same cloud)."""
    centres = np.stack([-(f.w2c[:3, :3].T @ f.w2c[:3, 3]) for f in frames])   # camera centre = -R^T t of the world-to-camera pose
    lo, hi = float(centres.min()), float(centres.max())
    mid, half = 0.5 * (lo + hi), (hi - lo) / 6.0
    xyzs = (mid - half) + np.random.rand(num_points, 3) * (2.0 * half)
    return Pointcloud(xyzs, np.full((num_points, 3), 127, dtype=np.uint8))


def load_blender_data(path: str, use_masks: bool, mask_expand_pixels: int, eval: bool, eval_in_val: bool, eval_in_test: bool,
                      white_background: bool) -> Tuple[List[Frame], Pointcloud, List[int], List[int]]:
    """nerf_synthetic layout (/root/reference/scene/blender_loader.py:74-114): the held-out frames (val and / or test, as asked)
    come FIRST in the frame list, the training split behind them; with `eval` off every frame trains.  The start cloud is
    generated from the cameras that train."""
    root = Path(path)
    read = lambda split: load_frames(root / f"transforms_{split}.json", use_masks, mask_expand_pixels, white_background)   # noqa: E731
    held_out = [fr for split, wanted in (("val", eval_in_val), ("test", eval_in_test)) if wanted for fr in read(split)]
    frames = held_out + read("train")
    n_held = len(held_out)
    eval_indexes = list(range(n_held))
    train_indexes = list(range(n_held if eval else 0, len(frames)))
    return frames, generate_pointcloud([frames[i] for i in train_indexes]), train_indexes, eval_indexes


# ------------------------------------------------------------------------------------------------ Scene
class SceneDataset(torch.utils.data.Dataset):
    def __init__(self, scene: "Scene", split: Literal["train", "eval"]):
        super().__init__()
        self.scene, self.split = scene, split

    def __len__(self):
        return self.scene.nbr_data(self.split)

    def __getitem__(self, idx):
        return self.scene.get_data(self.split, idx)


class Scene:
    def __init__(self, data_path: str, data_format: Literal["colmap", "blender"], output_path: Optional[str], total_iterations: int,
                 eval: bool, eval_split_ratio: float, eval_in_val: bool, eval_in_test: bool, use_masks: bool,
                 mask_expand_pixels: int, white_background: bool):
        if data_format == "colmap":
            loaded = load_colmap_data(data_path, use_masks, mask_expand_pixels, eval, eval_split_ratio, white_background)
        elif data_format == "blender":
            loaded = load_blender_data(data_path, use_masks, mask_expand_pixels, eval, eval_in_val, eval_in_test, white_background)
        else:
            raise ValueError(f"Invalid data_format: {data_format}")
        self.frames, self.pc, self.train_indexes, self.eval_indexes = loaded
        if total_iterations < len(self.train_indexes):
            raise ValueError("the number of iterations is less than the number of training data")
        self.train_indexes *= total_iterations // len(self.train_indexes) + 1
        self.train_indexes = self.train_indexes[:total_iterations]
        self.train_dataset = SceneDataset(self, "train")
        self.eval_dataset = SceneDataset(self, "eval")
        if output_path is not None:
            self._export_cameras_json(Path(output_path) / "cameras.json")

    def nbr_data(self, split: Literal["train", "eval"]) -> int:
        if split == "train":
            return len(self.train_indexes)
        if split == "eval":
            return len(self.eval_indexes)
        raise ValueError(f"Invalid split: {split}")

    def get_data(self, split: Literal["train", "eval"], index: int) -> Dict[str, Any]:
        if split == "train":
            return self.frames[self.train_indexes[index]].to_data()
        if split == "eval":
            return self.frames[self.eval_indexes[index]].to_data()
        raise ValueError(f"Invalid split: {split}")

    def _export_cameras_json(self, save_path: Path):
        with open(save_path, "w") as f:
            json.dump([frame.to_json(id) for id, frame in enumerate(self.frames)], f)
