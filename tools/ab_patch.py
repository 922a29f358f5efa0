"""Same-box A/B of MODULE CONSTANTS (the fused forms are constants, not environment switches: INTEGRATION.md section 5).

    python tools/ab_patch.py functional.PAD_WEIGHTS=False networks.pspnet_combine.FUSED_TAIL=False -- bench.py --steps 20 --warmup 5

Every ``module.NAME=value`` (module relative to the package, value a Python literal) is set before ``bench.py`` (or any script) runs
in this process under its own name; the JSON line gains nothing -- the caller labels the leg."""
import ast
import importlib
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "structure_knowledge_distillation_amd"


def main():
    args = sys.argv[1:]
    cut = args.index("--")
    for item in args[:cut]:
        path, value = item.split("=", 1)
        mod, name = path.rsplit(".", 1)
        m = importlib.import_module(PKG + "." + mod)
        if not hasattr(m, name):
            raise SystemExit("ab_patch: %s has no attribute %s" % (m.__name__, name))
        setattr(m, name, ast.literal_eval(value))
    script = args[cut + 1]
    sys.argv = [script] + args[cut + 2:]
    runpy.run_path(os.path.join(ROOT, script) if not os.path.isabs(script) else script, run_name="__main__")


if __name__ == "__main__":
    main()
