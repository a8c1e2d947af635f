"""Import shims so the pure-PyTorch reference (/root/reference) imports in the build container.

Build-time tooling only: used by tests/golden/make_golden.py to GENERATE the committed golden
vectors.  Nothing here (nor /root/reference) is needed at test / bench / smoke time.
Recipe follows SURVEY.md Appendix A.
"""
import importlib.abc
import importlib.machinery
import sys
import types
import typing

REFERENCE_ROOT = "/root/reference"

_MISSING_ROOTS = {
    "jaxtyping", "viser", "cv2", "nerfacc", "tensorboard", "torchvision", "git", "torchmetrics", "tyro",
    "wandb", "comet_ml", "mediapy", "open3d", "pymeshlab", "xatlas", "trimesh", "gsplat", "pytorch_msssim",
    "timm", "h5py", "zod", "vod", "numba", "pyquaternion", "nuscenes", "av2", "pandaset", "splines",
    "msgpack_numpy", "imageio", "skimage", "tinycudann",
}


class _Dummy:
    """Permissive stand-in: subscriptable, callable, iterable, attribute-bottomless."""

    def __init__(self, *a, **k):
        pass

    def __class_getitem__(cls, item):
        return cls

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # behaves as a pass-through decorator
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __iter__(self):
        return iter(())

    def __getitem__(self, item):
        return _Dummy()

    def __or__(self, other):
        return _Dummy

    __ror__ = __or__


class _StubModule(types.ModuleType):
    __version__ = "0.15.2"
    __path__ = []  # looks like a package so that submodule imports resolve

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


class _StubFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _MISSING_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
        return None


def install():
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.append(_StubFinder())
    tb = _StubModule("torch.utils.tensorboard")
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    git = _StubModule("git")
    git.Optional = typing.Optional
    sys.modules.setdefault("git", git)
