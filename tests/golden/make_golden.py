"""Generate the golden vectors under tests/golden/ by running the REFERENCE
itself (/root/reference/utilis.py, yolo_v3.py) on seeded inputs.

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

The reference imports cv2, torchvision, tensorboardX and skimage at module
level for code that is not on the inference path; none of them is installed
here, so empty stand-in modules are registered before the import.  Nothing
from them is called by parse_cfg / create_module / forward / predict /
box_iou / filter_img_class / filter_results.

Inputs come from gen_inputs.py (frozen RandomState streams); only outputs are
stored.  Large outputs are stored as (K, sha256, first/last rows).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import gen_inputs as gi  # noqa: E402
from yolo_amd.cfg import write_yolov3_cfg, default_params  # noqa: E402


def import_reference():
    for name in ["cv2", "tensorboardX", "torchvision", "torchvision.transforms",
                 "skimage", "skimage.color"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["skimage"].color = sys.modules["skimage.color"]
    sys.path.insert(0, "/root/reference")
    import utilis as ref_utilis
    import yolo_v3 as ref_yolo
    return ref_utilis, ref_yolo


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build_ref_model(ref_utilis, ref_yolo, params, cfg_path, seed):
    """Construct the reference model without a weights file: load_weights is
    made a no-op for the constructor, then the synthetic state_dict is loaded
    through the reference's own load_state_dict."""
    blocks = ref_utilis.parse_cfg(cfg_path)
    orig = ref_yolo.yolo_v3.load_weights
    ref_yolo.yolo_v3.load_weights = lambda self, *a, **k: None
    try:
        model = ref_yolo.yolo_v3(params, blocks)
    finally:
        ref_yolo.yolo_v3.load_weights = orig
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import yolo_oracle as orc
    table = orc.layer_table(orc.parse_cfg(cfg_path), params["channels"],
                            params["num_classes"], params["num_anchors"])
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in
          gi.synthetic_state_dict(table, seed).items()}
    model.load_state_dict(sd)
    model.eval()
    return model, blocks


def ref_filter(ref_utilis, pred, conf, nms, C, stable=False):
    params = {"confidence": conf, "nms_thesh": nms, "num_classes": C}
    t = torch.from_numpy(pred.copy())
    if stable:
        orig_sort = torch.sort
        def stable_sort(x, *a, **k):
            k["stable"] = True
            return orig_sort(x, *a, **k)
        torch.sort = stable_sort
        try:
            out = ref_utilis.filter_results(t, params)
        finally:
            torch.sort = orig_sort
    else:
        out = ref_utilis.filter_results(t, params)
    assert np.array_equal(t.numpy(), pred, equal_nan=True), "reference mutated its input"
    if isinstance(out, int):
        return out
    return out.numpy().astype(np.float32)


def main():
    ref_utilis, ref_yolo = import_reference()
    meta = {"torch": torch.__version__, "numpy": np.__version__}
    arrays = {}

    # ---- 7. parse_cfg / create_module ---------------------------------
    cfg_path = write_yolov3_cfg(os.path.join(HERE, "_tmp_yolov3.cfg"))
    params = default_params(416, 80, 3)
    blocks = ref_utilis.parse_cfg(cfg_path)
    for l in (81, 93, 105):
        blocks[l]["filters"] = 255
    ltd, module_list = ref_utilis.create_module(params, blocks)
    conv_table = []
    for i, m in enumerate(module_list):
        for sub in m.children():
            if isinstance(sub, torch.nn.Conv2d):
                conv_table.append([i, sub.in_channels, sub.out_channels,
                                   sub.kernel_size[0], sub.stride[0], sub.padding[0],
                                   sub.bias is not None, len(list(m.children()))])
    meta["layer_type_dic"] = {k: (v if k != "referred_relationship" else
                                   {str(a): (list(b) if isinstance(b, tuple) else b)
                                    for a, b in v.items()})
                              for k, v in ltd.items()}
    meta["conv_table"] = conv_table
    meta["num_parameters"] = int(sum(p.numel() for p in module_list.parameters()))
    meta["state_dict_keys_first"] = list(module_list.state_dict().keys())[:8]
    meta["n_blocks"] = len(blocks)
    meta["params_width_after_create"] = params["width"]

    # ---- 1. predict -----------------------------------------------------
    class _Layer(list):
        pass
    dec_cases = []
    for name, H, C, A, B in [("dec_h64", 64, 80, 3, 2), ("dec_h320_a1c4", 320, 4, 1, 1)]:
        p = default_params(H, C, A)
        p["cuda"] = False
        holder = types.SimpleNamespace(params=p)
        outs = []
        for k, s in enumerate((32, 16, 8)):
            G = H // s
            raw = gi.head_raw(100 + 10 * k + H, B, A, C, G)
            layer = [ref_utilis.yolo_layer(p["anchors"][k * A:(k + 1) * A])]
            o = ref_yolo.yolo_v3.predict(holder, torch.from_numpy(raw.copy()), layer, B)
            outs.append(o.numpy())
        arrays[name] = np.concatenate(outs, 1).astype(np.float32)
        dec_cases.append(dict(name=name, H=H, C=C, A=A, B=B))
    # exp overflow case: tw = 100 -> inf
    p = default_params(64, 80, 3); p["cuda"] = False
    raw = gi.head_raw(7, 1, 3, 80, 2)
    raw[0, 2, 0, 0] = 100.0
    raw[0, 85 + 3, 1, 1] = 95.0
    o = ref_yolo.yolo_v3.predict(types.SimpleNamespace(params=p), torch.from_numpy(raw.copy()),
                                 [ref_utilis.yolo_layer(p["anchors"][0:3])], 1)
    arrays["dec_overflow"] = o.numpy().astype(np.float32)
    meta["decode_cases"] = dec_cases

    # ---- 2. box_iou -------------------------------------------------------
    a, b = gi.iou_pairs(11)
    arrays["iou_pairs"] = ref_utilis.box_iou(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    arrays["iou_bcast"] = ref_utilis.box_iou(torch.from_numpy(a[100:101]),
                                             torch.from_numpy(b)).numpy()

    # ---- 3..5. filter_results --------------------------------------------
    nms_cases = []
    def add_nms(name, gen_kwargs, conf, nms, stable, store_full=True, edit=None):
        pred = gi.nms_pred(**gen_kwargs)
        if edit:
            pred = gi.EDITS[edit](pred)
        out = ref_filter(ref_utilis, pred, conf, nms, gen_kwargs["C"], stable=stable)
        case = dict(name=name, gen=gen_kwargs, conf=conf, nms=nms, stable=stable, edit=edit)
        if isinstance(out, int):
            case["K"] = 0
            case["sentinel_int"] = out
        else:
            case["K"] = int(out.shape[0])
            case["sha256"] = sha(out)
            if store_full:
                arrays["nms_" + name] = out
            else:
                case["first_rows"] = out[:3].tolist()
                case["last_rows"] = out[-3:].tolist()
                # per-image, per-class kept counts pin the structure
                case["rows_per_image"] = np.bincount(out[:, 0].astype(int),
                                                     minlength=pred.shape[0]).tolist()
        nms_cases.append(case)
        print(name, "K =", case["K"], flush=True)

    add_nms("uniform_n3000_c80_conf50", dict(seed=21, B=2, N=3000, C=80), 0.5, 0.4, False)
    add_nms("uniform_n3000_c80_conf01", dict(seed=22, B=2, N=3000, C=80), 0.01, 0.4, False)
    add_nms("clustered_n3000_c80_conf50", dict(seed=23, B=2, N=3000, C=80, mode="clustered"), 0.5, 0.4, False)
    add_nms("clustered_n600_c4_conf30", dict(seed=24, B=3, N=600, C=4, H=320, mode="clustered"), 0.3, 0.4, False)
    add_nms("uniform_n600_c4_conf01", dict(seed=25, B=3, N=600, C=4, H=320), 0.01, 0.45, False)
    add_nms("clustered_c1", dict(seed=26, B=2, N=500, C=1, mode="clustered"), 0.2, 0.4, False)
    # ties: objectness on a coarse grid, groups far beyond 16 -> stable sort
    add_nms("ties_uniform", dict(seed=31, B=2, N=3000, C=80, tie_levels=64), 0.5, 0.4, True)
    add_nms("ties_clustered", dict(seed=32, B=2, N=2000, C=4, mode="clustered", tie_levels=32), 0.25, 0.4, True)
    add_nms("ties_all_equal", dict(seed=33, B=1, N=700, C=3, mode="clustered", tie_levels=1,
                                   obj_lo=0.6, obj_hi=0.9), 0.5, 0.4, True)
    # sentinels and odd shapes
    add_nms("all_below", dict(seed=41, B=2, N=300, C=80, obj_hi=0.4), 0.5, 0.4, False)
    add_nms("single_box", dict(seed=42, B=2, N=300, C=80), 0.5, 0.4, False, edit="single_box")
    add_nms("middle_image_empty", dict(seed=43, B=3, N=300, C=80), 0.5, 0.4, False, edit="middle_empty")
    add_nms("exp_overflow", dict(seed=44, B=2, N=400, C=8, mode="clustered"), 0.3, 0.4, False, edit="overflow")
    add_nms("conf_equal_obj", dict(seed=45, B=1, N=200, C=5, tie_levels=4), 0.5, 0.4, True)
    add_nms("negative_conf", dict(seed=46, B=1, N=200, C=5), -1.0, 0.4, False, edit="some_zero_obj")
    # full-size (hash only)
    add_nms("full416_conf50", dict(seed=51, B=1, N=10647, C=80), 0.5, 0.4, False, store_full=False)
    add_nms("full416_conf01", dict(seed=52, B=1, N=10647, C=80), 0.01, 0.4, False, store_full=False)
    add_nms("full416_clustered", dict(seed=53, B=1, N=10647, C=80, mode="clustered"), 0.5, 0.4, False, store_full=False)
    add_nms("full608_conf50", dict(seed=54, B=1, N=22743, C=80, H=608), 0.5, 0.4, False, store_full=False)
    meta["nms_cases"] = nms_cases

    # ---- 6. forward -------------------------------------------------------
    fwd_cases = []
    for name, H, B, C, A in [("fwd_h64", 64, 2, 80, 3), ("fwd_h96", 96, 2, 80, 3),
                             ("fwd_h64_a1c4", 64, 1, 4, 1)]:
        p = default_params(H, C, A); p["cuda"] = False
        cfgp = write_yolov3_cfg(os.path.join(HERE, "_tmp_yolov3.cfg"), C, A)
        model, _ = build_ref_model(ref_utilis, ref_yolo, p, cfgp, seed=3)
        x = torch.from_numpy(gi.images(5, B, H))
        digests = {}
        hooks = []
        for i, m in enumerate(model.module_list):
            if i in model.layer_type_dic["conv"]:
                def hook(mod, inp, out, i=i):
                    o = out.detach().double()
                    flat = out.detach().reshape(-1)
                    idx = np.linspace(0, flat.numel() - 1, 8).astype(np.int64)
                    digests[str(i)] = dict(shape=list(out.shape), sum=float(o.sum()),
                                           abssum=float(o.abs().sum()),
                                           samples=flat[idx].tolist())
                hooks.append(m.register_forward_hook(hook))
        with torch.no_grad():
            y = model(x)
        for h in hooks:
            h.remove()
        y = y.numpy().astype(np.float32)
        case = dict(name=name, H=H, B=B, C=C, A=A, weight_seed=3, image_seed=5,
                    out_shape=list(y.shape), out_sum=float(y.astype(np.float64).sum()),
                    conv_digests=digests,
                    obj_min=float(y[:, :, 4].min()), obj_max=float(y[:, :, 4].max()),
                    frac_obj_gt_half=float((y[:, :, 4] > 0.5).mean()))
        if H == 64:
            arrays[name] = y
        else:
            idx = np.linspace(0, y.size - 1, 64).astype(np.int64)
            case["out_samples_idx"] = idx.tolist()
            case["out_samples"] = y.reshape(-1)[idx].tolist()
        # end to end through the reference's NMS as well
        out = ref_filter(ref_utilis, y, 0.5, 0.4, C, stable=False)
        if isinstance(out, int):
            case["e2e_K"] = 0
        else:
            case["e2e_K"] = int(out.shape[0])
            arrays[name + "_e2e"] = out
        fwd_cases.append(case)
        print(name, "obj range", case["obj_min"], case["obj_max"], "frac>0.5",
              case["frac_obj_gt_half"], "e2e K", case["e2e_K"], flush=True)
    meta["forward_cases"] = fwd_cases
    os.remove(os.path.join(HERE, "_tmp_yolov3.cfg"))

    np.savez_compressed(os.path.join(HERE, "golden_arrays.npz"), **arrays)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    total = sum(a.nbytes for a in arrays.values())
    print("wrote", len(arrays), "arrays,", total / 1e6, "MB raw")


if __name__ == "__main__":
    main()
