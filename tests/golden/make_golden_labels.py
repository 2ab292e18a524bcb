"""Pin prep_labels against the REFERENCE's own (/root/reference/utilis.py:519-535).

Run in the build container only:   python tests/golden/make_golden_labels.py

Seeded Darknet label files are written under a scratch directory (label_files()
below, shared with the test), the reference's prep_labels turns them into a csv
(whole set, and restricted to two classes), and the csv text goes to
golden_labels.json.  glob order depends on the file system, so the test compares
the data lines as a sorted list.
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import gen_inputs as gi  # noqa: E402

NAME_LIST = ["img_name", "c", "x", "y", "w", "h"]


def label_files(directory, seed=61, n_files=5):
    """n_files Darknet label files `txt_<k>.txt` (the doubled 'txt' shows that every
    occurrence is replaced), 0-4 lines each: class, then four fractions."""
    r = gi.rs(seed)
    os.makedirs(directory, exist_ok=True)
    for k in range(n_files):
        with open(os.path.join(directory, f"txt_{k}.txt"), "w") as fh:
            for _ in range(int(r.randint(0, 5))):
                fh.write("%d %.6f %.6f %.6f %.6f\n" % (r.randint(0, 4), r.uniform(), r.uniform(),
                                                       r.uniform(0.01, 0.5), r.uniform(0.01, 0.5)))


def run(prep_labels):
    """csv texts produced by `prep_labels` for the whole set and for classes 1 and 3
    (run from inside the scratch directory, so the stored paths are relative)."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            label_files("labels")
            prep_labels("labels/*.txt", NAME_LIST, "all.csv")
            prep_labels("labels/*.txt", NAME_LIST, "sel.csv", selected_cls=["1", "3"])
            prep_labels("labels/none_*.txt", NAME_LIST, "empty.csv")
            for name in ("all", "sel", "empty"):
                out[name] = open(name + ".csv").read()
        finally:
            os.chdir(cwd)
    return out


def main():
    from make_golden import import_reference
    ref_utilis, _ = import_reference()
    got = run(ref_utilis.prep_labels)
    with open(os.path.join(HERE, "golden_labels.json"), "w") as fh:
        json.dump(got, fh, indent=1)
    print(got["all"])
    print(got["sel"])
    print(repr(got["empty"]))


if __name__ == "__main__":
    main()
