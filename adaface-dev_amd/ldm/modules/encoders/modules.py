"""Host-side mirror of the reference's hooked CLIP text encoder, ``ldm/modules/encoders/modules.py``: ``FrozenCLIPEmbedder``
(:365-475) with the four forwards it monkey-patches into transformers' ``CLIPTextModel`` (``embeddings_forward`` :180-208,
``encoder_forward`` :212-259, ``text_model_forward`` :264-341, ``transformer_forward`` :345-363) -- SURVEY.md 8f rank 2.

What those hooks add to a plain CLIP text forward, and where it lives here:

* the embedding-manager call between the token-table lookup and the position add (``embedding_manager(input_ids, inputs_embeds)``
  replaces the embeddings of the subject placeholder tokens)            -> ``forward`` below;
* ``last_layers_skip_weights``: the final LayerNorm is applied to the weighted sum of the last k encoder states (k = 2, weights
  0.5 / 0.5 in every reference config; optionally re-drawn from a Dirichlet per call)  -> ``CLIPTextModelWrapper``'s
  ``hidden_state_layer_weights`` path (same arithmetic: normalised weights, sum in fp32, then ``final_layer_norm``);
* position embeddings extended from 77 to ``max_length`` by repeating the last rows  -> ``extend_position_embeddings``.

The 12 transformer layers run on the gfx950 kernels through the C ABI (``CLIPTextModelWrapper``: fused QKV GEMM, causal flash
attention, quick-GELU MLP epilogues, LayerNorm); gradients flow to the patched token embeddings through the per-op autograd nodes,
the encoder weights are frozen.  There is no transformers object underneath, so nothing is monkey-patched: ``initialize_hooks`` is a
no-op kept for call compatibility (reference ddpm.py:710).

Offline there are no CLIP vocabulary files: ``tokenizer`` can be any transformers-protocol tokenizer
(``CLIPTokenizer.from_pretrained(local_dir)``); the fallback is the deterministic ``WordTokenizer`` stand-in."""
import numpy as np
import torch
import torch.nn as nn

from ....adaface.arc2face_models import CLIPTextModelWrapper, clip_text_config


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class FrozenCLIPEmbedder(AbstractEncoder):
    """Uses the CLIP transformer encoder for text; same constructor arguments as the reference plus injectable parts."""

    def __init__(self, version="openai/clip-vit-large-patch14", device="cpu", max_length=77, last_layers_skip_weights=(0.5, 0.5),
                 randomize_clip_skip_weights=False, tokenizer=None, transformer=None, clip_config=None):
        super().__init__()
        if tokenizer is None:
            from ....adaface.adaface_wrapper import WordTokenizer
            tokenizer = WordTokenizer((clip_config or clip_text_config()).vocab_size)
        self.tokenizer = tokenizer
        self.transformer = transformer if transformer is not None else CLIPTextModelWrapper(clip_config or clip_text_config())
        if max_length != 77:
            self.transformer.extend_position_embeddings(max_length)
        self.device = device
        self.max_length = max_length
        self.set_last_layers_skip_weights(last_layers_skip_weights, use_as_dirichlet_weights=randomize_clip_skip_weights)

    def initialize_hooks(self):
        """The reference rebinds four transformers forwards here; this mirror implements them directly."""

    # NOTE: the last element is the weight of the last layer.
    def set_last_layers_skip_weights(self, weights, use_as_dirichlet_weights=False):
        if weights is None:
            self.transformer.text_model.last_layers_skip_weights = None
            self.dir_sampler = None
        elif not use_as_dirichlet_weights:
            w = np.array(weights, dtype=np.float64)
            self.transformer.text_model.last_layers_skip_weights = w / w.sum()
            self.dir_sampler = None
        else:
            self.dir_sampler = torch.distributions.dirichlet.Dirichlet(torch.tensor(list(weights), dtype=float))
            self.sample_last_layers_skip_weights()

    def sample_last_layers_skip_weights(self, verbose=False):
        if self.dir_sampler is None:
            return
        self.transformer.text_model.last_layers_skip_weights = self.dir_sampler.sample().numpy()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for param in self.parameters():
            param.requires_grad = False

    def tokenize(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True, return_overflowing_tokens=False,
                             padding="max_length", return_tensors="pt")
        return enc["input_ids"]

    def forward(self, text, embedding_manager=None, **kwargs):
        """text: list of prompts -> [B, max_length, 768].  ``embedding_manager`` patches the token embeddings (reference :192-196)."""
        if kwargs:
            raise NotImplementedError(f"unsupported arguments {sorted(kwargs)} (the AdaFace callers pass embedding_manager only)")
        tm = self.transformer.text_model
        dev = tm.final_layer_norm.weight.device
        tokens = self.tokenize(text).to(dev)
        with torch.no_grad():
            inputs_embeds = tm.embeddings.token_embedding(tokens)
        if embedding_manager is not None:
            inputs_embeds = embedding_manager(tokens, inputs_embeds)
        w = tm.last_layers_skip_weights
        w = None if w is None else torch.as_tensor(np.asarray(w), dtype=torch.float32, device=dev).view(-1, 1)
        return self.transformer(input_ids=tokens, input_token_embs=inputs_embeds, hidden_state_layer_weights=w)[0]

    def encode(self, text, **kwargs):
        return self(text, **kwargs)
