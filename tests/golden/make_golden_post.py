"""Golden vectors for the callers' rescale-and-clamp block
(/root/reference/detection.py:52-70, also video.py:78-91).

The block is inline in detection() (after the data loader and the model call), so
it cannot be imported and called; this script lets torch execute the same sequence
of tensor statements on seeded inputs and stores inputs and outputs in
golden_post.npz.  Run in the build container:  python tests/golden/make_golden_post.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def reference_block(output, im_dim_list, height):
    """The statements of detection.py:45,53-71 on CPU tensors."""
    im_dim_list = torch.FloatTensor(im_dim_list).repeat(1, 2)
    output = output.clone()
    im_dim_list = torch.index_select(im_dim_list, 0, output[:, 0].long())
    scaling_factor = torch.min(height / im_dim_list, 1)[0].view(-1, 1)
    output[:, [1, 3]] -= (height - scaling_factor * im_dim_list[:, 0].view(-1, 1)) / 2
    output[:, [2, 4]] -= (height - scaling_factor * im_dim_list[:, 1].view(-1, 1)) / 2
    output[:, 1:5] /= scaling_factor
    for i in range(output.shape[0]):
        output[i, [1, 3]] = torch.clamp(output[i, [1, 3]], 0.0, im_dim_list[i, 0])
        output[i, [2, 4]] = torch.clamp(output[i, [2, 4]], 0.0, im_dim_list[i, 1])
    return output


def case(seed, K, B, height, weird=False):
    r = np.random.RandomState(seed)
    rows = np.zeros((K, 9), dtype=np.float32)
    rows[:, 0] = r.randint(0, B, K)
    c = r.uniform(-0.2 * height, 1.2 * height, (K, 2))
    wh = r.uniform(1, 0.6 * height, (K, 2))
    rows[:, 1:3] = c - wh / 2
    rows[:, 3:5] = c + wh / 2
    rows[:, 5:7] = r.uniform(0, 1, (K, 2))
    rows[:, 7] = r.randint(0, 80, K)
    rows[:, 8] = rows[:, 2]
    dims = np.stack([r.randint(17, 4000, B), r.randint(17, 3000, B)], 1).astype(np.float32)
    if weird:
        rows[0, 1] = np.nan
        rows[1, 4] = np.inf
        rows[2, 2] = -np.inf
        dims[0] = (height, height)
        dims[1] = (1.0, 5000.0)
    return rows, dims


def main():
    arrays = {}
    for name, (seed, K, B, height, weird) in {
        "post_416": (21, 700, 8, 416, False),
        "post_608_weird": (22, 300, 5, 608, True),
        "post_320_one": (23, 1, 1, 320, False),
    }.items():
        rows, dims = case(seed, K, B, height, weird)
        out = reference_block(torch.from_numpy(rows), dims.tolist(), height).numpy()
        arrays[name + "_rows"] = rows
        arrays[name + "_dims"] = dims
        arrays[name + "_height"] = np.asarray(height)
        arrays[name + "_out"] = out
    np.savez_compressed(os.path.join(HERE, "golden_post.npz"), **arrays)
    print("wrote golden_post.npz:", {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main()
