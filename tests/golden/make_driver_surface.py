#!/usr/bin/env python3
"""Generate tests/golden/driver_surface.json: what the reference's driver scripts need from the modules this repository mirrors.

Build container only (reads /root/reference; the JSON it writes is data and is what travels).  For each driver
(main_image.py, main_vtab.py, main_video.py, speed.py) and for the reference's own engine_finetune.py the script
`ast`-parses the source and records

* ``imports``: every ``(module, attribute)`` pair the file imports or touches (``from misc import save_on_master``,
  ``misc.init_distributed_mode(...)``, ``lr_sched.adjust_learning_rate``, ...) from the modules on the hot path's boundary:
  ``misc``, ``engine_finetune``, ``block_flops_dict``, ``models`` / ``models.*``, ``video_models.*``, ``util.lr_sched``,
  ``util.metrics``  -> these must resolve in dynamic-tuning_amd/;
* ``reference_owned``: the ``(module, attribute)`` pairs from sibling modules that stay the reference's (``util.pos_embed``,
  ``util.logger``, ``configs``, ``datasets.*``, ``video_datasets.*``) -> these must NOT be shadowed by dynamic-tuning_amd/;
* ``calls``: the call sites of the boundary's functions with their positional-argument count and keyword names
  (``train_one_epoch(model, criterion, loader, optimizer, device, epoch, loss_scaler, max_norm=..., log_writer=..., args=..., logger=...)``)
  -> each must bind to the mirror's signature;
* ``model_attrs``: attributes / methods the drivers use on the model object the factory returns.

tests/test_driver_surface.py resolves every entry against dynamic-tuning_amd/ (CPU, no reference needed) and, where
/root/reference exists, imports the real drivers with dynamic-tuning_amd/ in front of the reference root.
Usage:  python tests/golden/make_driver_surface.py
"""
import ast
import json
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DRIVERS = ["main_image.py", "main_vtab.py", "main_video.py", "speed.py", "engine_finetune.py"]

OURS = ("misc", "engine_finetune", "block_flops_dict", "models", "video_models", "util.lr_sched", "util.metrics")
THEIRS = ("util.pos_embed", "util.logger", "util.crop", "util.datasets", "util.lars", "util.lr_decay", "configs", "datasets", "video_datasets")
MODEL_FACTORIES = {"vit_base_patch16_224_in21k"}


def _owner(module):
    for group, names in (("ours", OURS), ("theirs", THEIRS)):
        for n in names:
            if module == n or module.startswith(n + "."):
                return group
    return None


def scan(path):
    tree = ast.parse(open(path).read(), path)
    alias = {}      # local name -> module it is bound to (import x as y / import x.y as z)
    symbol = {}     # local name -> (module, attribute) for from-imports
    pairs = {"ours": set(), "theirs": set()}
    calls = []
    model_names, model_attrs = set(), set()

    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                if _owner(a.name):
                    alias[a.asname or a.name.split(".")[0]] = a.name if a.asname else a.name.split(".")[0]
                    pairs[_owner(a.name)].add((a.name, None))
        elif isinstance(node, ast.ImportFrom) and node.module and _owner(node.module):
            for a in node.names:
                symbol[a.asname or a.name] = (node.module, a.name)
                pairs[_owner(node.module)].add((node.module, a.name))

    def dotted(n):
        parts = []
        while isinstance(n, ast.Attribute):
            parts.append(n.attr)
            n = n.value
        if isinstance(n, ast.Name):
            parts.append(n.id)
            return list(reversed(parts))
        return None

    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            d = dotted(node)
            if d and d[0] in alias and len(d) >= 2:
                mod = alias[d[0]]
                if _owner(mod):
                    pairs[_owner(mod)].add((mod, d[1]))
        if isinstance(node, ast.Call):
            d = dotted(node.func) if isinstance(node.func, ast.Attribute) else ([node.func.id] if isinstance(node.func, ast.Name) else None)
            target = None
            if d and len(d) == 1 and d[0] in symbol and _owner(symbol[d[0]][0]) == "ours":
                target = symbol[d[0]]
            elif d and len(d) == 2 and d[0] in alias and _owner(alias[d[0]]) == "ours":
                target = (alias[d[0]], d[1])
            if target:
                calls.append(dict(module=target[0], name=target[1], npos=len(node.args),
                                  keywords=sorted(k.arg for k in node.keywords if k.arg), line=node.lineno))
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and isinstance(node.value.func, ast.Name) \
                and node.value.func.id in MODEL_FACTORIES:
            for t in node.targets:
                if isinstance(t, ast.Name):
                    model_names.add(t.id)
    model_names |= {"model_without_ddp"} if model_names else set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            d = dotted(node)
            if d and d[0] in model_names and len(d) >= 2 and d[1] != "module":
                model_attrs.add(".".join(d[1:3]) if d[1] == "head" and len(d) > 2 else d[1])
    order = lambda ps: sorted(([m, a] for m, a in ps), key=lambda e: (e[0], e[1] or ""))   # attribute None = `import module`
    return dict(imports=order(pairs["ours"]), reference_owned=order(pairs["theirs"]),
                calls=sorted(calls, key=lambda c: c["line"]), model_attrs=sorted(model_attrs))


def main():
    out = {"generated_by": "tests/golden/make_driver_surface.py", "reference_files": {}}
    for name in DRIVERS:
        out["reference_files"][name] = scan(os.path.join(REF, name))
    dst = os.path.join(HERE, "driver_surface.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    n = sum(len(v["imports"]) for v in out["reference_files"].values())
    print("wrote %s: %d (module, attribute) pairs over %d files" % (dst, n, len(DRIVERS)))


if __name__ == "__main__":
    main()
