"""Per-layer conv timing of one bench step from a rocprofv3 kernel trace CSV."""
import csv, glob, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from yolo_amd import engine
from yolo_amd.cfg import yolov3_cfg_text
import yolo_oracle as orc

def main(path, B=32, H=416):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    fused = any('conv_stem' in n for n in names)
    idx = [i for i, n in enumerate(names) if ('conv_stem' in n if fused else 'conv_first' in n)][-1]
    table = engine.graph_from_blocks(orc.parse_cfg_text(yolov3_cfg_text()), 3, 80, 3)
    ops, heads, info = engine.build_ops(table, B, H, 3)
    if fused:                      # first two layers are one launch
        merged = dict(ops[1]); merged['stem'] = ops[0]
        ops = [merged] + ops[2:]
    # a split reduction is two launches: the finishing kernel's time is added to its layer's
    convs = []
    for r in rows[idx:]:
        if 'splitk_finish' in r['Kernel_Name'] and convs:
            convs[-1] = dict(convs[-1], End_Timestamp=r['End_Timestamp'], split=True)
        elif 'conv' in r['Kernel_Name']:
            if len(convs) == len(ops):
                break
            convs.append(dict(r))
    tot = 0; groups = {}
    prev_end = None; gaps = 0
    for op, r in zip(ops, convs):
        e = op['ent']
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if prev_end: gaps += (int(r['Start_Timestamp']) - prev_end) / 1e3
        prev_end = int(r['End_Timestamp'])
        fl = 2 * B * op['Ho'] * op['Wo'] * e['cout'] * e['cin'] * e['k'] ** 2
        if 'stem' in op:
            o0 = op['stem']; e0 = o0['ent']
            fl += 2 * B * o0['Ho'] * o0['Wo'] * e0['cout'] * e0['cin'] * e0['k'] ** 2
        key = (e['k'], e['stride'], e['cin'], e['cout'], op['Ho'])
        key = key + (('split',) if r.get('split') else ())
        g = groups.setdefault(key, [0, 0.0, 0.0]); g[0] += 1; g[1] += dur; g[2] += fl
        tot += dur
    print(f"{'k s cin cout Ho':34s} {'n':>3s} {'us each':>9s} {'us total':>9s} {'TF/s':>7s}")
    for key, (n, d, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print(f"{str(key):34s} {n:3d} {d/n:9.1f} {d:9.1f} {fl/d/1e6:7.0f}")
    print("conv kernels total us", round(tot, 1), "gaps us", round(gaps, 1))

if __name__ == "__main__":
    main(sys.argv[1], *[int(v) for v in sys.argv[2:4]])
