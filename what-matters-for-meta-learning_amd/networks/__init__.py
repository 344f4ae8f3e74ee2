"""Drop-in `networks` package: same module / class names, constructor signature,
forward signature and state_dict layout as the reference's networks/ (train.py:41-45),
with the arithmetic running in hand-written HIP kernels (libmlhot.so)."""

import importlib.abc
import importlib.machinery
import sys
import types

# Methods of the reference that are OUTSIDE the accelerated hot path (MAML-style inner-loop adaptation and the
# single-task baselines).  `importlib.import_module(f"networks.{method}")` (train.py:41) still resolves for them - to a module whose
# class refuses construction with a clear message - instead of eight near-identical stub files.
OUT_OF_SCOPE = {
    "MAMLMR": "MAML inner-loop adaptation", "MAMLMRShapeNet1D": "MAML inner-loop adaptation", "MAMLShapeNet1D": "MAML inner-loop adaptation",
    "MMAMLShapeNet1D": "MAML inner-loop adaptation", "VanillaMAML": "MAML inner-loop adaptation",
    "SingleTaskDistractor": "single-task baseline", "SingleTaskShapeNet1D": "single-task baseline", "SingleTaskShapeNet3D": "single-task baseline",
}


def _refusing_class(name, what):
    from torch import nn

    def __init__(self, config=None, *args, **kwargs):
        raise NotImplementedError(f"method '{name}' ({what}) is not part of the MI355X hot-path build; "
                                  "in scope: CNP*/ANP* (vanilla, ResNet, MR and Distractor variants) - see INTEGRATION.md")
    return type(name, (nn.Module,), {"__init__": __init__, "__doc__": f"`networks.{name}` of the reference: {what}, out of scope here."})


class _OutOfScopeFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        pkg, _, name = fullname.rpartition(".")
        if pkg == __name__ and name in OUT_OF_SCOPE:
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return types.ModuleType(spec.name)

    def exec_module(self, module):
        name = module.__name__.rpartition(".")[2]
        setattr(module, name, _refusing_class(name, OUT_OF_SCOPE[name]))


if not any(isinstance(f, _OutOfScopeFinder) for f in sys.meta_path):
    sys.meta_path.append(_OutOfScopeFinder())
