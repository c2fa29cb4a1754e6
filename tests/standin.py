"""Stand-in eps-model shared by tests/golden/gen_golden.py (where it drives the REFERENCE's host code: ``UNetTeacher.forward``,
``LatentDiffusion.guided_denoise``, ``calc_unet_distill_loss``) and by the tests (where the same function drives the oracle and
the mirrors).  It is a fixed cheap nonlinear map of (x, t, context) whose output also encodes the per-call flags the reference
routes through ``extra_info``, so a fixture generated through it pins the ORCHESTRATION (which instances / contexts / flags /
timesteps each U-Net call gets, how results are combined), not U-Net arithmetic.  Weights come from adaface_dev_amd.rng, so both
sides rebuild it from the seed."""
import torch
import torch.nn.functional as F


class StandInEps(torch.nn.Module):
    def __init__(self, ctx_dim=16, seed=61):
        super().__init__()
        from adaface_dev_amd import rng
        self.register_buffer("w", rng.synth_input("standin.w", (4, 4, 3, 3), seed=seed) * 0.3)
        self.register_buffer("p", rng.synth_input("standin.p", (ctx_dim, 4), seed=seed) * 0.5)

    def forward(self, x, t, ctx):
        """x [B,4,h,w], t [B] int, ctx [B,T,D] -> eps [B,4,h,w] (differentiable in x and ctx)."""
        w, p = self.w.to(x.dtype), self.p.to(x.dtype)
        h = F.conv2d(x, w, padding=1) + (ctx.to(x.dtype).mean(dim=1) @ p)[:, :, None, None]
        h = h + torch.sin(t.to(x.dtype) * 0.01)[:, None, None, None]
        return torch.tanh(h) * 0.8 + 0.1 * x


FLAG_WEIGHTS = (("normalize_cross_attn", 0.05), ("mix_attn_mats_in_batch", 0.1), ("use_attn_lora", 0.2), ("use_ffn_lora", 0.4))


class StandInWrapper(torch.nn.Module):
    """Plays ``LatentDiffusion.model`` (the U-Net wrapper, reference ddpm.py:4187-4252): ``model(x, t, cond_context)``.
    The flags found in ``extra_info`` shift eps by fixed amounts, ``img_mask`` scales it, and with ``capture_ca_activations`` a small
    activation dict (one tensor, one nested dict, one list entry per instance) is left in ``extra_info['ca_layers_activations']`` like
    the reference wrapper does (:4233-4239).  ``calls`` records (batch size, flags) per call."""

    def __init__(self, eps_model):
        super().__init__()
        self.eps_model = eps_model
        self.calls = []

    def forward(self, x, t, cond_context, out_dtype=torch.float32):
        ctx, prompts, extra = cond_context
        eps = self.eps_model(x, t, ctx)
        code = 0.0
        for name, wgt in FLAG_WEIGHTS:
            if bool(extra.get(name, False)):
                code += wgt
        if extra.get("ffn_lora_adapter_name") == "unet_distill":
            code += 0.8
        eps = eps + code
        if extra.get("img_mask") is not None:
            eps = eps * (0.5 + 0.5 * extra["img_mask"].to(eps.dtype))
        gs = extra.get("res_hidden_states_gradscale", 1)
        if gs != 1:
            eps = eps * 1.0 + 0.001 * gs
        self.calls.append((x.shape[0], {k: (bool(extra.get(k, False))) for k, _ in FLAG_WEIGHTS}, extra.get("ffn_lora_adapter_name"),
                           len(prompts), torch.is_grad_enabled()))
        if extra.get("capture_ca_activations", False):
            extra["ca_layers_activations"] = {"outfeat": {23: eps * 2.0, 24: eps[:, :2] + 1.0}, "attn": eps.mean(dim=(2, 3)),
                                              "names": [f"inst{i}" for i in range(x.shape[0])]}
        else:
            extra["ca_layers_activations"] = {"outfeat": {}, "attn": eps.new_zeros(x.shape[0], 0), "names": []}
        return eps.to(out_dtype)


# ----------------------------------------------------------------------------- stand-ins for the CLIP text encoders of the ID -> prompt stack
CLIP_WORD_IDS = {"photo": 1125, "of": 539, "a": 320, ",": 267, "id": 1014, "person": 2533}
CLIP_BOS, CLIP_EOS = 49406, 49407


class WordTokenizer:
    """Call protocol of the CLIP tokenizer for the few template prompts of the ID -> prompt stack (whitespace split, the six words
    above; the vocabulary files of the real tokenizer are not available offline).  ',' glued to a word ("a ,") is split off."""

    def encode(self, text, add_special_tokens=False):
        return [CLIP_WORD_IDS[w] for w in text.replace(",", " , ").split()]

    def __call__(self, prompts, truncation=True, padding="max_length", max_length=77, return_tensors="pt"):
        prompts = [prompts] if isinstance(prompts, str) else list(prompts)
        rows = []
        for p in prompts:
            ids = [CLIP_BOS] + self.encode(p) + [CLIP_EOS]
            ids = ids[:max_length] + [CLIP_EOS] * (max_length - len(ids))
            rows.append(ids)
        import types
        return types.SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))


class StandInCLIP(torch.nn.Module):
    """Plays ``CLIPTextModelWrapper`` (adaface/arc2face_models.py:236-338) in the fixtures that pin the GLUE around it
    (slot replacement, layer weights, static suffix, slicing, averaging / perturbation stages): same call protocol --
    ``(input_ids=, return_token_embs=True)`` -> token embeddings, ``(input_ids=, input_token_embs=, hidden_state_layer_weights=)`` ->
    ``(hidden [B, T, D],)`` -- computed by a fixed cheap nonlinear map that depends on every argument (differentiable in the token
    embeddings and the layer weights)."""

    def __init__(self, hidden_size=768, seed=73):
        super().__init__()
        import types
        from adaface_dev_amd import rng
        self.config = types.SimpleNamespace(hidden_size=hidden_size)
        self.table = torch.nn.Parameter(rng.synth_input("standin.clip.table", (1024, hidden_size), seed=seed), requires_grad=False)
        self.pos = torch.nn.Parameter(rng.synth_input("standin.clip.pos", (128, hidden_size), seed=seed) * 0.1, requires_grad=False)
        self.w = torch.nn.Parameter(rng.synth_input("standin.clip.w", (hidden_size, hidden_size), seed=seed) * hidden_size ** -0.5, requires_grad=False)

    def forward(self, input_ids=None, input_token_embs=None, hidden_state_layer_weights=None, return_token_embs=False, **kw):
        if return_token_embs:
            return self.table[input_ids % 1024].clone()
        x = self.table[input_ids % 1024] if input_token_embs is None else input_token_embs
        x = x.to(self.w.dtype)
        h = torch.tanh((x + self.pos[: x.shape[1]]) @ self.w)
        if hidden_state_layer_weights is not None:
            wl = hidden_state_layer_weights.to(h.dtype).reshape(-1)
            h = sum(wl[k] / wl.sum() * (h * (0.5 + 0.25 * k) + 0.05 * k) for k in range(wl.numel()))
        return (h,)


# ----------------------------------------------------------------------------- stand-ins around the Stage-2 / recon loss assemblies
class StandInCaptureWrapper(StandInWrapper):
    """``StandInWrapper`` whose captured-activation dict has every key the Stage-2 losses read, for layers 22-24, as fixed cheap
    differentiable maps of (eps, context): q / q2 / attn_out [B, C, N], outfeat [B, C, h, w], k / v [B, C, L], attn / attnscore
    [B, heads, N, L] (rows softmaxed over the L context tokens)."""
    C, HEADS = 8, 2

    def __init__(self, eps_model, ctx_dim=16, seed=63):
        super().__init__(eps_model)
        from adaface_dev_amd import rng
        for li in (22, 23, 24):
            self.register_buffer(f"pf{li}", rng.synth_input(f"standin.cap.pf{li}", (self.C, 4), seed=seed) * 0.7)
            self.register_buffer(f"pq{li}", rng.synth_input(f"standin.cap.pq{li}", (self.C, 4), seed=seed) * 0.9)
            self.register_buffer(f"pk{li}", rng.synth_input(f"standin.cap.pk{li}", (ctx_dim, self.C), seed=seed) * 0.5)
            self.register_buffer(f"pv{li}", rng.synth_input(f"standin.cap.pv{li}", (ctx_dim, self.C), seed=seed) * 0.5)

    def forward(self, x, t, cond_context, out_dtype=torch.float32):
        eps = super().forward(x, t, cond_context, out_dtype)
        ctx, _, extra = cond_context
        if extra.get("capture_ca_activations", False):
            B, _, h, w = eps.shape
            acts = {k: {} for k in ("outfeat", "attn", "attnscore", "q", "q2", "k", "v", "attn_out")}
            for li in (22, 23, 24):
                pf, pq, pk, pv = (getattr(self, f"p{n}{li}").to(eps.dtype) for n in "fqkv")
                feat = torch.einsum("cd,bdhw->bchw", pf, eps)
                q = torch.einsum("cd,bdhw->bchw", pq, torch.tanh(eps + 0.1 * li - 2.3)).reshape(B, self.C, h * w)
                k, v = (ctx.to(eps.dtype) @ pk).permute(0, 2, 1), (ctx.to(eps.dtype) @ pv).permute(0, 2, 1)
                d = self.C // self.HEADS
                score = torch.einsum("bhdn,bhdl->bhnl", q.reshape(B, self.HEADS, d, h * w), k.reshape(B, self.HEADS, d, -1))
                attn = score.softmax(dim=-1)
                ao = torch.einsum("bhnl,bhdl->bhdn", attn, v.reshape(B, self.HEADS, d, -1)).reshape(B, self.C, h * w)
                acts["outfeat"][li], acts["attn_out"][li] = feat + 0.3 * ao.reshape(B, self.C, h, w), ao
                acts["q"][li], acts["q2"][li], acts["k"][li], acts["v"][li] = q, q, k, v
                acts["attn"][li], acts["attnscore"][li] = attn, score
            extra["ca_layers_activations"] = acts
        return eps


FACE_PRESETS = ((40, 24, 56, 64), (16, 40, 64, 56), (56, 48, 48, 48), (24, 16, 72, 80))      # (x, y, w, h) on a 128 x 128 image


def standin_detect(image_np, T=20):
    """One face per image whose box is picked by coarse image content (left / right and top / bottom brightness), so that the reference
    side and the mirror side of a fixture -- which decode the same latents with the same stand-in decoder -- see the same boxes."""
    g = image_np.astype("float64").mean(axis=2)
    H, W = g.shape
    i = int(g[:, : W // 2].mean() > g[:, W // 2:].mean()) * 2 + int(g[: H // 2].mean() > g[H // 2:].mean())
    x, y, w, h = FACE_PRESETS[i]
    sx, sy = W / 128.0, H / 128.0
    return [(x * sx, y * sy, w * sx, h * sy, 0.995)]


def standin_detect_small_second_face(image_np, T=20):
    """As standin_detect plus a smaller second (background) face."""
    return standin_detect(image_np, T) + [(4.0, 4.0, 30.0, 28.0, 0.93)]


def standin_detect_dark_images_faceless(image_np, T=20):
    """As standin_detect, but an image whose mean brightness is below 100 shows no face.  In the recon scenario (stage2_scenario.recon_inputs) the
    second instance's x0 prediction of the first denoising step is such an image (mean 75 against ~126 for the others): a PARTIALLY detected
    batch, which is where the instance weights of the recon loss matter (reference ddpm.py:2738-2739: a LONG mask, so the 0.1 meant for
    instances without a face truncates to 0)."""
    return [] if image_np.astype("float64").mean() < 100.0 else standin_detect(image_np, T)


def standin_decode(z):
    """Plays the VAE decoder: latent [B, 4, h, w] -> image [B, 3, 8h, 8w] in about [-1, 1], differentiable."""
    return F.interpolate(torch.tanh(z[:, :3] * 0.9 + 0.2 * z[:, 3:4]), scale_factor=8, mode="bilinear", align_corners=False)


class StandInFaceNet(torch.nn.Module):
    """Plays ResNetFace-18 behind ArcFaceWrapper: grey [N, 1, 128, 128] -> [N, 24] embeddings, differentiable, fp32."""

    def __init__(self, seed=64):
        super().__init__()
        from adaface_dev_amd import rng
        self.w1 = torch.nn.Parameter(rng.synth_input("standin.face.w1", (6, 1, 5, 5), seed=seed) * 0.4, requires_grad=False)
        self.w2 = torch.nn.Parameter(rng.synth_input("standin.face.w2", (24, 6 * 16), seed=seed) * 0.3, requires_grad=False)

    def forward(self, grey):
        h = torch.tanh(F.conv2d(grey.float(), self.w1, stride=4, padding=2))
        return F.adaptive_avg_pool2d(h, 4).flatten(1) @ self.w2.t()
