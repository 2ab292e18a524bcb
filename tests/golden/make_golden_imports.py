"""Which names the reference's four inference callers import from the modules this
package replaces.  Run in the build container only:

    python tests/golden/make_golden_imports.py

The callers' source is parsed (ast), nothing is executed; only the module and
symbol NAMES are stored, in golden_imports.json.  The test requires every one
of them to resolve through dropin/ (get_fps.py asks for `LetterboxImage_cv`,
a name the reference's utilis.py does not define: its own import fails; the
drop-in exports it as an alias).
"""
import ast
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
CALLERS = ["detection.py", "video.py", "get_fps.py", "evaluate.py"]
REPLACED = {"utilis", "yolo_v3", "evaluate", "generate_anchors"}


def main():
    out = {}
    for caller in CALLERS:
        tree = ast.parse(open(os.path.join(REF, caller)).read())
        names = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module in REPLACED:
                names.setdefault(node.module, set()).update(a.name for a in node.names)
            elif isinstance(node, ast.Import):
                for a in node.names:
                    if a.name in REPLACED:
                        names.setdefault(a.name, set())
        out[caller] = {m: sorted(v) for m, v in sorted(names.items())}
    # attributes the callers use on the model object (grep of `model.` in the four files)
    out["model_attributes"] = ["cuda", "eval", "load_weights", "params", "parameters", "train"]
    with open(os.path.join(HERE, "golden_imports.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
