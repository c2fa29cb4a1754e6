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
