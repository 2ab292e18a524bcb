"""Host logic of the drop-in modules, no GPU: cfg parsing, module construction
(names, routing table, state_dict keys) against what the reference produced,
the fused execution plan, the Darknet .weights reader."""
import os
import tempfile

import numpy as np
import pytest
import torch

import gen_inputs as gi
import yolo_oracle as orc
from helpers import load_golden
from yolo_amd.cfg import default_params, write_yolov3_cfg, yolov3_cfg_text
from yolo_amd import engine
from yolo_amd.utilis import compute_ap, create_module, parse_cfg
from yolo_amd.yolo_v3 import yolo_v3

META, ARR = load_golden()


@pytest.fixture(scope="module")
def cfg_path():
    path = write_yolov3_cfg()
    yield path
    os.remove(path)


def test_parse_cfg_and_create_module_match_the_reference(cfg_path):
    blocks = parse_cfg(cfg_path)
    assert len(blocks) == 107 and all(isinstance(v, str) for b in blocks for v in b.values())
    assert blocks == orc.parse_cfg(cfg_path)
    params = default_params(416)
    del params["width"]
    for l in (81, 93, 105):
        blocks[l]["filters"] = 255
    ltd, modules = create_module(params, blocks)
    assert params["width"] == 416
    want = META["layer_type_dic"]
    for key in ("conv", "upsampling", "shortcut", "route_1", "route_2", "yolo", "referred"):
        assert list(ltd[key]) == want[key], key
    rel = {str(k): (list(v) if isinstance(v, tuple) else v) for k, v in ltd["referred_relationship"].items()}
    assert rel == want["referred_relationship"]
    assert list(modules.state_dict().keys())[:8] == META["state_dict_keys_first"]
    assert sum(p.numel() for p in modules.parameters()) == META["num_parameters"]
    convs = []
    for i, m in enumerate(modules):
        for sub in m.children():
            if isinstance(sub, torch.nn.Conv2d):
                convs.append([i, sub.in_channels, sub.out_channels, sub.kernel_size[0], sub.stride[0],
                              sub.padding[0], sub.bias is not None, len(list(m.children()))])
    assert convs == META["conv_table"]
    # first head gets anchors[0:A] (largest first)
    assert np.array_equal(modules[82][0].anchors, params["anchors"][0:3])
    assert np.array_equal(modules[106][0].anchors, params["anchors"][6:9])


def test_malformed_route_raises_like_the_reference():
    text = yolov3_cfg_text().replace("layers=-1, 61", "layers=-1, 61, 36")
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as fh:
        fh.write(text)
    with pytest.raises(Exception, match="Route layer"):
        create_module(default_params(416), parse_cfg(fh.name))
    os.remove(fh.name)


def test_constructor_needs_weights_like_the_reference(cfg_path):
    params = default_params(416)
    with pytest.raises(FileNotFoundError):
        yolo_v3(params, parse_cfg(cfg_path))        # default ../4TrainingWeights/yolov3.weights


def test_snapshot_constructor_and_cpu_forward_is_refused(cfg_path):
    table = orc.layer_table(orc.parse_cfg(cfg_path), 3, 80, 3)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in gi.synthetic_state_dict(table, 3).items()}
    with tempfile.NamedTemporaryFile(suffix=".pth", delete=False) as fh:
        torch.save(sd, fh.name)
    params = default_params(64)
    params["pretrain_snapshot"] = fh.name
    model = yolo_v3(params, parse_cfg(cfg_path))
    os.remove(fh.name)
    assert model.params is params and len(model.module_list) == 107
    got = model.state_dict()
    assert set(got) == set(sd)
    assert torch.equal(got["module_list.81.conv_81.bias"], sd["module_list.81.conv_81.bias"])
    model.eval()
    if not torch.cuda.is_available():
        with pytest.raises(Exception, match="CPU"):
            model(torch.zeros(1, 3, 64, 64))          # no CPU fallback, fails loudly
    with pytest.raises(NotImplementedError):
        model(torch.zeros(1, 3, 64, 64), is_training=True)


@pytest.mark.parametrize("B,H", [(32, 416), (1, 320), (32, 608), (2, 64)])
def test_fused_plan_structure_and_buffer_placement(B, H):
    blocks = orc.parse_cfg_text(yolov3_cfg_text())
    table = engine.graph_from_blocks(blocks, 3, 80, 3)
    ops, heads, info = engine.build_ops(table, B, H, 3)
    assert len(ops) == 75 and sum(1 for o in ops if o["res"]) == 23
    assert [g for _, g in heads] == [H // 32, H // 16, H // 8]
    assert [o["index"] for o in ops if o["fp32"]] == [81, 93, 105]
    up = [o for o in ops if len(o["src"]) == 2]
    assert [(o["index"], o["src"][0][2], o["src"][0][1], o["src"][1][1]) for o in up] == \
        [(87, 1, 256, 512), (99, 1, 128, 256)]
    assert up[0]["src"][1][0] == "L60" and up[1]["src"][1][0] == "L35"   # shortcut outputs 61 / 36
    flops = sum(2 * B * o["Ho"] * o["Wo"] * o["ent"]["cout"] * o["ent"]["cin"] * o["ent"]["k"] ** 2
                for o in ops)
    if H == 416:
        assert abs(flops / B / 1e9 - 65.864) < 0.01
    if H == 608:
        assert abs(flops / B / 1e9 - 140.692) < 0.01
    offsets, peak = engine.place_buffers(info)
    spans = [(offsets[b], offsets[b] + v["bytes"], v["first"], v["last"], b) for b, v in info.items()]
    for i, (o1, e1, f1, l1, b1) in enumerate(spans):
        assert o1 % 256 == 0
        for (o2, e2, f2, l2, b2) in spans[i + 1:]:
            overlap_time = not (l1 < f2 or l2 < f1)
            overlap_mem = not (e1 <= o2 or e2 <= o1)
            assert not (overlap_time and overlap_mem), (b1, b2)
    assert peak < 0.3 * sum(v["bytes"] for v in info.values())
    # an op never writes over something it is reading
    for o in ops:
        reads = [s[0] for s in o["src"] if s[0] != "input"] + ([o["res"]] if o["res"] else [])
        d0 = offsets[f"L{o['index']}"]
        d1 = d0 + info[f"L{o['index']}"]["bytes"]
        for r in reads:
            assert d1 <= offsets[r] or offsets[r] + info[r]["bytes"] <= d0


def test_outputs_of_a_fused_group_never_land_on_the_groups_inputs():
    """Launches that run several layers (the stem with the 1x1 layer behind it, a 3x3 layer with
    the 1x1 layer behind it) write the later layers' outputs while the first layer's inputs are
    still read by other workgroups.  Without the lifetime extension the placement puts block 6's
    output exactly there: on block 4's output, which dies with block 5."""
    blocks = orc.parse_cfg_text(yolov3_cfg_text())
    table = engine.graph_from_blocks(blocks, 3, 80, 3)

    def group_hazards(extend):
        ops, _, info = engine.build_ops(table, 32, 416, 3)
        fuse = [0] * len(ops)
        fuse[0] = 2                                    # blocks 0, 1, 2
        for j, o in enumerate(ops):
            if o["index"] in (5, 7):                   # 3x3 64 -> 128 with the 1x1 128 -> 64 behind it
                fuse[j] = 1
        if extend:
            engine.extend_fused_lifetimes(ops, info, fuse)
        offsets, _ = engine.place_buffers(info)
        bad = []
        j = 0
        while j < len(ops):
            n = 1 + fuse[j]
            group = ops[j:j + n]
            outs = {f"L{o['index']}" for o in group}
            reads = {s[0] for o in group for s in o["src"] if s[0] != "input"} | {o["res"] for o in group if o["res"]}
            for r in reads - outs:
                for w in outs:
                    if not (offsets[w] + info[w]["bytes"] <= offsets[r] or offsets[r] + info[r]["bytes"] <= offsets[w]):
                        bad.append((w, r))
            j += n
        return bad

    assert group_hazards(True) == []
    assert ("L6", "L3") in group_hazards(False), "the case this guards against no longer arises: pick another"


def test_unsupported_graphs_are_rejected_not_mis_executed():
    blocks = orc.parse_cfg_text(yolov3_cfg_text())
    blocks[85]["stride"] = "4"
    table = engine.graph_from_blocks(blocks, 3, 80, 3)
    with pytest.raises(engine.UnsupportedGraph):
        engine.build_ops(table, 1, 416, 3)


def test_compute_ap():
    rec = np.array([0.1, 0.2, 0.2, 0.4, 0.7, 1.0])
    prec = np.array([1.0, 1.0, 0.66, 0.75, 0.6, 0.5])
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    idx = np.where(mrec[1:] != mrec[:-1])[0]
    assert compute_ap(rec, prec) == pytest.approx(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


def test_dropin_directory_resolves_the_reference_module_names():
    """With dropin/ first on sys.path the reference callers' imports (`from utilis
    import ...`, `from evaluate import get_map`, `from generate_anchors import
    set_anchors_to_model`) land in this package."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r);"
            "import utilis, evaluate, generate_anchors, yolo_v3;"
            "print(evaluate.get_map.__module__, generate_anchors.kmeans.__module__,"
            " utilis.filter_results.__module__, yolo_v3.yolo_v3.__module__)") % os.path.join(root, "dropin")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["yolo_amd.evaluate", "yolo_amd.generate_anchors",
                                  "yolo_amd.utilis", "yolo_amd.yolo_v3"]
