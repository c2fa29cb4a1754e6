"""Inputs shared by tests/golden/gen_golden.py::gen_embedding_manager (which drives the REFERENCE EmbeddingManager) and
tests/test_embedding_manager.py (which drives this package's mirror): a fake ID->prompt encoder, a text-embedder shell around the
WordTokenizer, and the prompt cases.  No reference code here."""
import types

import torch
import torch.nn as nn

E = 64                                     # embedding width of the fixtures (the bookkeeping does not depend on it)
FILLERS = lambda k: ", " * k


class FakeID2AdaPromptEncoder(nn.Module):
    """generate_adaface_embeddings(img_prompt_embs [b,K,E]) -> 2 x + 1: enough to see which embedding landed where."""
    name = "arc2face"

    def __init__(self, num_id_vecs=16, num_static_img_suffix_embs=0):
        super().__init__()
        self.num_id_vecs = num_id_vecs
        self.num_static_img_suffix_embs = num_static_img_suffix_embs
        self.subj_basis_generator = nn.Linear(2, 2)
        self.calls = []

    def generate_adaface_embeddings(self, image_paths=None, face_id_embs=None, img_prompt_embs=None, p_dropout=0,
                                    return_zero_embs_for_dropped_encoders=True, avg_at_stage=None, perturb_at_stage=None,
                                    perturb_std=0, enable_static_img_suffix_embs=None):
        self.calls.append(dict(bs=img_prompt_embs.shape[0], p_dropout=p_dropout, sfx=enable_static_img_suffix_embs))
        return img_prompt_embs * 2 + 1, None, [img_prompt_embs.shape[1]]

    def load_adaface_ckpt(self, path):
        self.loaded = path


def text_embedder(tokenizer, table):
    """The two attributes EmbeddingManager.__init__ reads from FrozenCLIPEmbedder."""
    emb = lambda tokens: table[tokens]
    return types.SimpleNamespace(tokenizer=tokenizer, transformer=types.SimpleNamespace(
        text_model=types.SimpleNamespace(embeddings=emb)))


def token_table(vocab=49408):
    g = torch.Generator().manual_seed(11)
    return torch.randn(vocab, E, generator=g)


def id_embs(bs, K, seed):
    return torch.randn(bs, K, E, generator=torch.Generator().manual_seed(seed))


# name -> (iter_type, subject names, prompts, id-emb batch, K, real_batch_size, training?)
CASES = {
    "distill": ("unet_distill_iter", ["alice", "bob"],
                ["a photo of z" + FILLERS(15) + " in a park", "portrait of a z" + FILLERS(15)], 2, 16, 2, False),
    "compos": ("compos_distill_iter", ["alice"],
               ["a z" + FILLERS(3) + " riding a horse", "a z" + FILLERS(3) + " on the moon",
                "a young woman" + FILLERS(3) + " riding a horse", "a young woman" + FILLERS(3) + " on the moon"], 1, 4, 1, True),
    "gap_and_repeat": ("recon_iter", ["default"],
                       ["z , dog , , , a z again", "the z , , cat , ,"], 2, 4, 2, False),
    "plain": ("plain_text_iter", ["default"], ["a photo of a dog", ""], None, 16, 2, False),
    "perturbed": ("unet_distill_iter", ["bob"], ["z" + FILLERS(3), "face of z" + FILLERS(3)], 2, 4, 2, True),
}


def flatten_indices(p2i):
    out = {}
    for k, v in p2i.items():
        out[k] = None if v is None else torch.stack([v[0], v[1]]).cpu()
    return out
