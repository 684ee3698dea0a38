"""The data side of the path: the 3DPW test stream (reference ``boa_dataset/pw3d.py``) and the Human3.6M exemplar set the
retrieval step draws from (reference ``base_adaptor.py:451-506`` ``SourceDataset``), in the reference's on-disk formats -
per-sequence ``data/dataset_extras/3dpw_<seq>_<pid>.npz`` (keys imgname, scale, center, pose, shape, j2d, op_j2d, gender)
and the joblib file ``h36m_random_sample_center_10_10.pt`` (imgname, scale, center, pose, shape, S, part[, gender]).

What differs from the reference: the crop / anti-aliased resize / normalisation of every frame runs on the GPU
(``dyb_crop_resize_normalize``, csrc/preprocess.hip) from the decoded uint8 frame instead of NumPy + skimage in dataloader
workers; frames are decoded by PIL (cv2 is not a dependency) on a small thread pool that runs ahead of the consumer.
Box corners and keypoint transforms are the reference's integer arithmetic (``utils/dataprocess.py:12-46``), on the host.
Items and batches carry the reference's keys and shapes, already on the device."""
from __future__ import annotations

import glob
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, constants as C
from ._abi import check
from .hmr import stream_of

PW3D_ROOT = "/data/syguan/human_datasets/3dpw"                                   # reference config.py:7-8
H36M_ROOT = "/data/syguan/human_datasets/Human3.6M/human36m_full_raw"
DATASET_NPZ_PATH = "data/dataset_extras"


# ---- box / keypoint arithmetic (reference utils/dataprocess.py:12-46, rot = 0) ---------------------------------------
def get_transform(center, scale, res):
    h = 200 * scale
    t = np.zeros((3, 3))
    t[0, 0] = float(res[1]) / h
    t[1, 1] = float(res[0]) / h
    t[0, 2] = res[1] * (-float(center[0]) / h + .5)
    t[1, 2] = res[0] * (-float(center[1]) / h + .5)
    t[2, 2] = 1
    return t


def transform(pt, center, scale, res, invert=0):
    t = get_transform(center, scale, res)
    if invert:
        t = np.linalg.inv(t)
    new_pt = np.dot(t, np.array([pt[0] - 1, pt[1] - 1, 1.]).T)
    return new_pt[:2].astype(int) + 1


def crop_box(center, scale, res=(C.IMG_RES, C.IMG_RES)):
    """(ul, br): corners of the crop box in frame pixels (utils/dataprocess.py:51-54)."""
    ul = np.array(transform([1, 1], center, scale, res, invert=1)) - 1
    br = np.array(transform([res[0] + 1, res[1] + 1], center, scale, res, invert=1)) - 1
    return ul, br


def j2d_processing(kp: np.ndarray, center, scale, res: int = C.IMG_RES) -> np.ndarray:
    """Keypoints (n, 3: x, y, conf) into the crop frame, normalised to [-1, 1] (boa_dataset/pw3d.py:138-151, test time)."""
    kp = np.array(kp, dtype=np.float64)
    for i in range(kp.shape[0]):
        kp[i, 0:2] = transform(kp[i, 0:2] + 1, center, scale, [res, res])
    kp[:, :-1] = 2. * kp[:, :-1] / res - 1.
    return kp.astype("float32")


def read_image(path: str) -> np.ndarray:
    """Decoded frame as uint8 RGB (H, W, 3) (the reference: cv2.imread(...)[:, :, ::-1], pw3d.py:128)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"), dtype=np.uint8)


_CROP_WS: Dict[tuple, torch.Tensor] = {}


def preprocess_frame(img_u8: torch.Tensor, center, scale, res: int = C.IMG_RES, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 (H, W, 3) RGB device tensor -> normalised (3, res, res) fp32 crop (reference rgb_processing + Normalize)."""
    lib = _lib.load()
    img_u8 = img_u8.contiguous()
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3:
        raise ValueError("preprocess_frame expects a uint8 (H, W, 3) RGB frame")
    H, W = int(img_u8.shape[0]), int(img_u8.shape[1])
    ul, br = crop_box(center, scale, [res, res])
    bh, bw = int(br[1] - ul[1]), int(br[0] - ul[0])
    if bh <= 0 or bw <= 0:
        raise ValueError(f"empty crop box for center={center} scale={scale}")
    nbytes = int(lib.dyb_crop_workspace_bytes(bh, bw))
    dev = img_u8.device
    sid = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    key = (str(dev), sid)
    ws = _CROP_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _CROP_WS[key] = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.empty(3, res, res, dtype=torch.float32, device=dev)
    m, s = C.IMG_NORM_MEAN, C.IMG_NORM_STD
    check(lib.dyb_crop_resize_normalize(img_u8.data_ptr(), H, W, int(ul[0]), int(ul[1]), int(br[0]), int(br[1]), out.data_ptr(), res,
                                        m[0], m[1], m[2], s[0], s[1], s[2], ws.data_ptr(), ws.numel(), stream_of(img_u8)),
          "dyb_crop_resize_normalize")
    torch.autograd.graph.increment_version(out)     # written by a raw kernel: caches keyed on (storage, version) must see the new frame
    return out


def key_3dpw(elem: str) -> int:
    """Ordering of the sequence files (boa_dataset/pw3d.py:19-23)."""
    elem = os.path.basename(elem)
    return int(elem.split('_')[1]) * 10 + int(elem.split('_')[2][:-4])


def _genders(data, n) -> np.ndarray:
    try:
        return np.array([0 if str(g) == 'm' else 1 for g in data['gender']]).astype(np.int32)
    except KeyError:
        return -1 * np.ones(n).astype(np.int32)


class PW3D:
    """3DPW test set: every frame of every ``3dpw_<seq>_<pid>.npz`` in sequence order (reference PW3D, pw3d.py:26-196).
    ``__getitem__`` returns the reference's item (image, smpl_j2d, op_j2d, pose, betas, gender, imgname, dataset_name, j3d,
    bbox) with tensors on ``device``; ``sequences`` lists (file, first index, frame count) for sharding by sequence."""

    def __init__(self, options=None, npz_dir: str = DATASET_NPZ_PATH, img_dir: str = PW3D_ROOT, device="cuda", files: Optional[Sequence[str]] = None):
        self.options, self.img_dir, self.device = options, img_dir, torch.device(device)
        self.files = sorted(files if files is not None else glob.glob(os.path.join(npz_dir, '3dpw_[0-9]*_[0-9].npz')), key=key_3dpw)
        cols: Dict[str, list] = {k: [] for k in ("imgname", "scale", "center", "pose", "betas", "smpl_j2d", "op_j2d", "gender")}
        self.sequences: List[dict] = []
        first = 0
        for f in self.files:
            d = np.load(f)
            n = int(d['scale'].shape[0])
            cols["imgname"].append(d['imgname']); cols["scale"].append(d['scale']); cols["center"].append(d['center'])
            cols["pose"].append(d['pose'].astype(np.float64)); cols["betas"].append(d['shape'].astype(np.float64))
            cols["smpl_j2d"].append(d['j2d']); cols["op_j2d"].append(d['op_j2d']); cols["gender"].append(_genders(d, n))
            self.sequences.append(dict(file=f, first=first, frames=n))
            first += n
        cat = lambda k: np.concatenate(cols[k], axis=0) if cols[k] else np.zeros((0,))
        self.imgnames, self.scales, self.centers = cat("imgname"), cat("scale"), cat("center")
        self.pose, self.betas, self.smpl_j2ds, self.op_j2ds, self.genders = cat("pose"), cat("betas"), cat("smpl_j2d"), cat("op_j2d"), cat("gender")
        if options is not None and getattr(options, "expdir", None):
            d = os.path.join(options.expdir, options.expname)
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, 'seq_order.record'), 'w') as fh:            # pw3d.py:66-68
                fh.writelines(x + '\n' for x in self.files)

    def __len__(self):
        return int(self.scales.shape[0])

    def host_item(self, index: int) -> dict:
        """Everything of an item that is host work: the decoded frame and the transformed annotations."""
        scale, center = float(self.scales[index]), np.array(self.centers[index], dtype=np.float64)
        imgname = str(self.imgnames[index])
        return dict(frame=read_image(os.path.join(self.img_dir, imgname)), scale=scale, center=center, imgname=imgname,
                    op_j2d=j2d_processing(self.op_j2ds[index], center, scale), smpl_j2d=j2d_processing(self.smpl_j2ds[index], center, scale),
                    pose=self.pose[index].astype("float32"), betas=self.betas[index].astype("float32"), gender=int(self.genders[index]))

    def device_item(self, h: dict) -> dict:
        dev = self.device
        frame = torch.from_numpy(h["frame"]).to(dev, non_blocking=True)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
        return dict(image=preprocess_frame(frame, h["center"], h["scale"]), op_j2d=t(h["op_j2d"]), smpl_j2d=t(h["smpl_j2d"]),
                    pose=t(h["pose"]), betas=t(h["betas"]), gender=torch.tensor(h["gender"], dtype=torch.long, device=dev),
                    imgname=h["imgname"], dataset_name='3dpw', j3d=torch.zeros(24, 4, device=dev),
                    bbox=torch.tensor([h["center"][0], h["center"][1], h["scale"] * 200], dtype=torch.float64, device=dev))

    def __getitem__(self, index: int) -> dict:
        return self.device_item(self.host_item(index))


def collate(items: Sequence[dict]) -> dict:
    """torch DataLoader default_collate semantics for the item dicts above."""
    out = {}
    for k in items[0]:
        v = [it[k] for it in items]
        out[k] = torch.stack(v, 0) if torch.is_tensor(v[0]) else list(v)
    return out


class FrameLoader:
    """``DataLoader(dataset, batch_size, shuffle=False, num_workers=8)`` of reference base_adaptor.py:137 for the datasets of
    this module: host work (decode + annotation transforms) on `workers` threads running `prefetch` batches ahead, the
    upload + crop kernel on the consumer's stream.  ``indices`` restricts / orders the walk (a rank's sequence shard)."""

    def __init__(self, dataset, batch_size: int = 1, workers: int = 8, prefetch: int = 4, indices: Optional[Sequence[int]] = None):
        self.ds, self.bs, self.workers, self.prefetch = dataset, batch_size, workers, prefetch
        self.indices = list(range(len(dataset))) if indices is None else list(indices)

    def __len__(self):
        return (len(self.indices) + self.bs - 1) // self.bs

    def __iter__(self):
        batches = [self.indices[i:i + self.bs] for i in range(0, len(self.indices), self.bs)]
        with ThreadPoolExecutor(max_workers=max(1, self.workers)) as pool:
            pending = []
            nxt = 0
            while nxt < len(batches) or pending:
                while nxt < len(batches) and len(pending) < self.prefetch:
                    pending.append([pool.submit(self.ds.host_item, i) for i in batches[nxt]])
                    nxt += 1
                futs = pending.pop(0)
                yield collate([self.ds.device_item(f.result()) for f in futs])


class SourceDataset:
    """The Human3.6M exemplar set behind ``BaseAdaptor.retrieval`` (reference base_adaptor.py:451-506): ``__getitem__`` ->
    {'keypoints' (1,49,3), 'img' (1,3,224,224), 'pose' (1,72), 'betas' (1,10), 'pose_3d' (1,24,4), 'imgname'} on `device`."""

    def __init__(self, datapath: str, img_dir: str = H36M_ROOT, device="cuda"):
        import joblib
        self.img_dir, self.device = img_dir, torch.device(device)
        d = self.data = joblib.load(datapath)
        self.imgname, self.scale, self.center = d['imgname'], d['scale'], d['center']
        self.pose, self.betas, self.pose_3d = d['pose'].astype(np.float64), d['shape'].astype(np.float64), d['S']
        kp_gt = d['part']
        self.keypoints = np.concatenate([np.zeros((len(self.imgname), 25, 3)), kp_gt], axis=1)      # OpenPose slots empty (:466-467)
        self.gender = _genders(d, len(self.imgname)) if 'gender' in d else -1 * np.ones(len(self.imgname)).astype(np.int32)
        self.length = self.scale.shape[0]

    def __len__(self):
        return len(self.imgname)

    def __getitem__(self, index: int) -> dict:
        dev = self.device
        scale, center = float(self.scale[index]), np.array(self.center[index], dtype=np.float64)
        imgname = os.path.join(self.img_dir, str(self.imgname[index]))
        frame = torch.from_numpy(read_image(imgname)).to(dev)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev).unsqueeze(0)
        return dict(keypoints=t(j2d_processing(self.keypoints[index], center, scale)),
                    img=preprocess_frame(frame, center, scale).unsqueeze(0), pose=t(self.pose[index]), betas=t(self.betas[index]),
                    imgname=imgname, pose_3d=t(self.pose_3d[index]))
