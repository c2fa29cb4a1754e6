"""Import machinery for tests/golden/gen_golden.py (build container only): makes the REFERENCE's modules importable
where their third-party dependencies are absent from this image (pytorch_lightning, diffusers, peft, muon, insightface,
cv2, ... -- SURVEY.md 8c) by fabricating EMPTY stand-in modules for exactly those packages.  A stand-in holds no
behaviour: every attribute is a do-nothing class, so only reference code whose arithmetic does not go through those
packages can be driven this way (the teacher loop, the attention-score rewrites, the loss functions) -- which is what
the golden vectors generated through it cover."""
import importlib.abc
import importlib.machinery
import sys
import types

import torch

ABSENT = ("pytorch_lightning", "muon", "diffusers", "peft", "gma", "cv2", "insightface", "bitsandbytes", "torchvision",
          "omegaconf", "ConsistentID", "kornia", "albumentations", "wandb", "onnxruntime", "retinaface", "facexlib",
          "open_clip", "clip", "taming", "easydict", "scipy_absent", "torchmetrics", "lpips", "scikit_image", "skimage", "deepface", "pytorch_fid")


class _Meta(type):
    def __getattr__(cls, name):                          # logging.get_logger(...), SomeEnum.VALUE ...: absorb
        if name.startswith("_"):
            raise AttributeError(name)
        return _Dummy


class _Dummy(torch.nn.Module, metaclass=_Meta):
    """Permissive placeholder: constructible with anything, usable as a base class or a pass-through decorator."""

    def __new__(cls, *a, **k):
        if cls is _Dummy and len(a) == 1 and not k and callable(a[0]) and not isinstance(a[0], type):
            return a[0]                                  # used as a decorator: hand the function back
        return super().__new__(cls)

    def __init__(self, *a, **k):
        super().__init__()



class _StubModule(types.ModuleType):
    __path__ = []                                        # a package: any submodule import resolves through the finder

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "ListConfig":
            v = type("ListConfig", (list,), {})
        elif name[:1].isupper() and name.isupper():      # CONSTANT_LIKE
            v = name
        else:
            v = type(name, (_Dummy,), {"__module__": self.__name__})
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def install(ref="/root/reference"):
    """transformers must be fully imported BEFORE the fake torchvision can be seen."""
    import transformers  # noqa: F401
    import transformers.models.clip.modeling_clip  # noqa: F401
    from transformers import CLIPTextModel, CLIPTokenizer, CLIPVisionModel  # noqa: F401
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.append(_Finder())
    if ref not in sys.path:
        sys.path.insert(0, ref)
