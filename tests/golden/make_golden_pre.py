"""Golden vectors for the pad / normalise / permute part of the callers' image
transforms: LetterBoxImage_cv after its cv2.resize call
(/root/reference/utilis.py:80-86), ImgToTensorCv (:107-116), the Normalize of
/root/reference/detection.py:34-35, and the tail of prep_image (:514-515).

cv2 is not installed, so the transforms cannot be called as a whole; this script
feeds an already-resized image to the statements that follow the resize -- executed
by numpy and torch, ImgToTensorCv by the reference's own class -- and stores the
outputs in golden_pre.npz (full arrays for small canvases, sha256 + samples for the
big one).  torchvision's Normalize is tensor.sub_(mean).div_(std) with float32
(C, 1, 1) tensors; torchvision is absent too, so that statement is written out.
Build container only:   python tests/golden/make_golden_pre.py
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_inputs as gi  # noqa: E402

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
CASES = {
    # name: (seed, canvas side, resized w, resized h, lo, hi)
    "landscape_416": (81, 416, 416, 234, 0, 256),
    "portrait_96": (82, 96, 40, 96, 0, 256),
    "full_64_narrow_range": (83, 64, 64, 64, 30, 200),
    "dark_32": (84, 32, 32, 20, 140, 250),       # the 128 padding is the minimum
}


def import_reference_utilis():
    for name in ["cv2", "tensorboardX", "torchvision", "torchvision.transforms",
                 "skimage", "skimage.color"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["skimage"].color = sys.modules["skimage.color"]
    sys.path.insert(0, "/root/reference")
    import utilis
    return utilis


def after_resize(resized_image, w, h, inp_channel=3):
    """utilis.py:80-86 with self.w, self.h = w, h."""
    new_h, new_w = resized_image.shape[:2]
    canvas = np.full((h, w, inp_channel), 128)
    canvas[(h-new_h)//2:(h-new_h)//2 + new_h, (w-new_w)//2:(w-new_w)//2 + new_w, :] = resized_image
    plain = canvas.copy()
    with np.errstate(divide="ignore", invalid="ignore"):
        canvas = (canvas - canvas.min())/(canvas.max()-canvas.min())
    return plain, canvas


def normalize(tensor, mean, std):
    tensor = tensor.clone()
    mean = torch.as_tensor(mean, dtype=tensor.dtype).view(-1, 1, 1)
    std = torch.as_tensor(std, dtype=tensor.dtype).view(-1, 1, 1)
    return tensor.sub_(mean).div_(std)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = import_reference_utilis()
    to_tensor = ref.ImgToTensorCv()
    arrays = {}
    for name, (seed, side, rw, rh, lo, hi) in CASES.items():
        resized = gi.photo(seed, rw, rh, lo, hi)
        plain, normed = after_resize(resized, side, side)
        t_minmax = to_tensor(normed)
        t_det = normalize(t_minmax, MEAN, STD)
        t_prep = torch.from_numpy(plain[:, :, ::-1].transpose((2, 0, 1)).copy()).float().div(255.0)
        for key, t in (("minmax", t_minmax), ("detection", t_det), ("prep", t_prep)):
            a = t.numpy()
            arrays[f"{name}_{key}_sha"] = np.array(sha(a))
            arrays[f"{name}_{key}_sample"] = a[:, ::max(1, side // 8), ::max(1, side // 8)].copy()
            if side <= 96:
                arrays[f"{name}_{key}"] = a
        print(name, float(t_det.min()), float(t_det.max()))
    flat = np.full((32, 32, 3), 128, dtype=np.uint8)
    _, normed = after_resize(flat, 32, 32)
    arrays["flat_32_minmax"] = to_tensor(normed).numpy()
    np.savez_compressed(os.path.join(HERE, "golden_pre.npz"), **arrays)
    print("wrote golden_pre.npz")


if __name__ == "__main__":
    main()
