"""Container-only helper: import the reference's Python files from /root/reference.

TEST INFRASTRUCTURE. Used only by tests/golden/make_golden.py (fixture generation, runs in the
authoring container where /root/reference is mounted).  Nothing here is imported by the product
package, and nothing here runs on the GPU box (the reference does not travel).

The reference does `from torchvision import models, transforms` at import time
(new_scripy.py:7-8, MNIST_script.py:24-26); torchvision is not installed in this image, so an empty
stand-in package is registered first.  No hot-path class touches torchvision.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("DM_REFERENCE_ROOT", "/root/reference")


def _stub_torchvision():
    if "torchvision" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    for sub in ("models", "transforms", "utils", "datasets"):
        m = types.ModuleType("torchvision." + sub)
        setattr(tv, sub, m)
        sys.modules["torchvision." + sub] = m
    sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
    sys.modules["torchvision.utils"].make_grid = lambda *a, **k: None
    sys.modules["torchvision.datasets"].MNIST = object
    sys.modules["torchvision"] = tv


def load(name):
    """Return the reference module `name` ("new_scripy" or "MNIST_script")."""
    _stub_torchvision()
    import matplotlib
    matplotlib.use("Agg")
    path = os.path.join(REF_ROOT, name + ".py")
    spec = importlib.util.spec_from_file_location("_dm_ref_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "new_scripy.py"))
