"""Reader for the reference's AdaFace checkpoints (``embeddings_gs-N.pt``, written by ``EmbeddingManager.save``,
embedding_manager.py:513-524): a dict whose ``string_to_subj_basis_generator_dict`` entry is a *pickled* ``nn.ModuleDict`` of
reference ``SubjBasisGenerator`` objects (class paths ``adaface.subj_basis_generator.*``, ``adaface.arc2face_models.*`` and
transformers' CLIP modules), plus ``placeholder_strings``, ``subject_strings`` and the ``unet_lora_modules`` state dict.

The reference packages are not importable next to this one, so the unpickler resolves their classes to this package's mirrors
(same attribute / parameter names, so ``state_dict()`` of the restored object has the reference's keys) and anything else that
cannot be imported to an attribute-bag ``nn.Module`` shell.  Objects are restored without running constructors (pickle protocol),
they are only used as containers of tensors and plain attributes (``N_ID``, ``prompt2token_proj_attention_multipliers`` ...)."""
import importlib
import pickle

import torch
import torch.nn as nn


class ShellModule(nn.Module):
    """Stands in for a pickled class that is not importable here: keeps whatever ``__dict__`` the pickle carries."""

    def forward(self, *a, **k):
        raise RuntimeError("checkpoint shell object: only state_dict() / attributes are usable")


def _mirror(module, name):
    from . import arc2face_models, face_id_to_ada_prompt, subj_basis_generator
    table = {"adaface.subj_basis_generator": subj_basis_generator, "adaface.arc2face_models": arc2face_models,
             "adaface.face_id_to_ada_prompt": face_id_to_ada_prompt,
             # older checkpoints were written when these modules lived under ldm.modules (embedding_manager.py:5-7)
             "ldm.modules.subj_basis_generator": subj_basis_generator, "ldm.modules.arc2face_models": arc2face_models}
    return getattr(table.get(module), name, None) if module in table else None


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        m = _mirror(module, name)
        if m is not None:
            return m
        if module.split(".")[0] in ("adaface", "ldm"):
            return ShellModule
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return ShellModule


class _PickleModule:
    """The ``pickle_module`` protocol torch.load expects."""
    __name__ = "adaface_ckpt_pickle"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump, dumps, Pickler = staticmethod(pickle.dump), staticmethod(pickle.dumps), pickle.Pickler
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL


def load_adaface_ckpt_file(path, map_location="cpu"):
    """torch.load of an AdaFace checkpoint with reference classes mapped as described above."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_PickleModule)


def module_state_dict(obj):
    """state dict of a restored module / shell, or the object itself if the checkpoint stored a plain state dict."""
    if isinstance(obj, dict) and not isinstance(obj, nn.Module):
        return obj
    return obj.state_dict()


# ----------------------------------------------------------------------------- writer (repo -> reference)
# attributes of this package's layers that hold derived device state (fp16 weight packs and their keys): never pickled
_CACHE_ATTRS = ("_cache", "_cache_bwd", "_pk", "_pack_key", "_packs", "_packs_key", "_prefix")

# class paths the REFERENCE environment resolves (transformers 4.x layout, reference requirements.txt:13; the reference's own
# modules under `adaface.`); the mirror's leaf layers are torch.nn layers with a HIP forward -> pickled as their torch base class
_TRANSFORMERS_CLIP = "transformers.models.clip.modeling_clip"
_REF_CLASS_BY_NAME = {
    "CLIPTextTransformer": _TRANSFORMERS_CLIP, "CLIPTextEmbeddings": _TRANSFORMERS_CLIP, "CLIPEncoder": _TRANSFORMERS_CLIP,
    "CLIPEncoderLayer": _TRANSFORMERS_CLIP, "CLIPMLP": _TRANSFORMERS_CLIP,
}


def reference_class_path(cls):
    """(module, qualname) under which the reference environment finds the counterpart of one of this package's classes
    (None: pickle the class under its own path)."""
    mod = cls.__module__
    if not mod.startswith("adaface_dev_amd."):
        return None
    if cls.__qualname__ in _REF_CLASS_BY_NAME and mod.endswith("adaface.arc2face_models"):
        return _REF_CLASS_BY_NAME[cls.__qualname__], cls.__qualname__
    for base in cls.__mro__[1:]:                      # HIP-backed leaf layer (Linear, LayerNorm, Conv2d ...): its torch.nn base
        if base.__module__.startswith("torch.nn.modules") and base is not nn.Module:
            return base.__module__, base.__qualname__
    rel = mod[len("adaface_dev_amd."):]
    if rel.startswith(("adaface.", "ldm.")):
        return rel, cls.__qualname__
    return None


class _RefPickler(pickle._Pickler):
    """Pure-Python pickler (so that class references can be rewritten) that emits the reference's class paths and drops the
    derived device caches from module state."""

    def save_global(self, obj, name=None):
        ref = reference_class_path(obj) if isinstance(obj, type) else None
        if ref is None:
            return super().save_global(obj, name)
        self.write(pickle.GLOBAL + ref[0].encode() + b"\n" + ref[1].encode() + b"\n")
        self.memoize(obj)

    def reducer_override(self, obj):
        if isinstance(obj, nn.Module) and type(obj).__module__.startswith("adaface_dev_amd."):
            import copyreg
            state = {k: v for k, v in obj.__dict__.items() if k not in _CACHE_ATTRS}
            return copyreg.__newobj__, (type(obj),), state
        return NotImplemented


class _RefPickleModule:
    __name__ = "adaface_ckpt_ref_pickle"
    Pickler = _RefPickler
    dump = staticmethod(lambda obj, f, protocol=None: _RefPickler(f, protocol).dump(obj))
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL


def save_adaface_ckpt_file(obj, path):
    """torch.save in the reference's on-disk form (embedding_manager.py:513-524): module objects are pickled whole, but under
    the class paths the REFERENCE resolves (``adaface.subj_basis_generator.SubjBasisGenerator``, transformers' CLIP modules,
    ``torch.nn`` layers) and without this package's derived state (fp16 weight packs), so the file round-trips both ways:
    the reference's ``torch.load(..., weights_only=False)`` restores its own classes, and ``load_adaface_ckpt_file`` maps them
    back to the mirrors.  Parameters are written as CPU tensors."""
    def cpu(o):
        if isinstance(o, nn.Module):
            import copy
            memo = {}
            for m in o.modules():                     # do not deep-copy device caches
                for a in _CACHE_ATTRS:
                    if a in m.__dict__:
                        memo[id(m.__dict__[a])] = None
            return copy.deepcopy(o, memo).to("cpu")
        if isinstance(o, dict):
            return type(o)((k, cpu(v)) for k, v in o.items())
        if torch.is_tensor(o):
            return o.detach().cpu()
        return o
    torch.save(cpu(obj), path, pickle_module=_RefPickleModule)
